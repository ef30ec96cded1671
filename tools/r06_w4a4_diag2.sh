#!/bin/bash
# Round-6 diagnosis of the W4A4 (FP6-pipe) loop, second pass: ring / window / pacing / priority variants of the shipped
# tuple-ring form interleaved (tuning library), in-k-step timelines of the FP6 and the int8 loop (consumer wave 0 = shares its SIMD with a loader
# wave, wave 2 = does not), short SQ / LDS / TA counter passes, then the GPU suite and the bench line.  Output: gpurun_out/r06e_* (small files only).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
C=wr128x192_s16_d4_l2,wr128x192_f6r_abl3_mfma,wr128x192_f6r_abl1_noW,wr128x192_f6r_abl2_noX,wr128x192_f6r_abl32_dma_noreads,wr128x192_f6r_abl33_reads_nodma,wr128x192_f6r_abl6_nobar,wr128x192_f6r_oldswz,wr128x192_f6r_xr8,wr128x192_f6r_xr2,wr128x192_f6r_pace,wr128x192_f6r_delay,wr128x192_f6r_xr8_pace,wr128x192_f6r_xr8_delay,wr128x192_f6r_lprio0,wr128x192_f6r_cprio3,wr128x192_f6r_xr8_lprio0,wr128x192_f6r_pace_lprio0,wr128x192_f6r_xr8_pace_lprio0,wr128x192_f6r_l4,wr128x192_f6r_d3,wr128x192_f6r_d4,wr128x192_f6r_s6,wr128x192_f6r_s10
MIXQ_TUNING_LIB=1 timeout 400 python3 tools/prefill_sweep.py --bit 4 --tokens 512 --layers 11008x4096 --cfgs $C --rounds 15 2>&1 | grep -v amdgpu.ids > $O/r06e_f6_variants.txt
tr -s ' ' < $O/r06e_f6_variants.txt | sed 's/ wr/\n  wr/g; s/ auto/\n  auto/g; s/ int_mm/\n  int_mm/' | head -60
timeout 300 python3 tools/trace_kstep.py --cfg wr128x192_f6r_t36_kstep_timeline,wr128x192_f6r_t_wave2,wr128x192_f6r_t_xr8,wr128x192_f6r_t_pace,wr128x192_f6r_t_xr8_pace 2>&1 | grep -v amdgpu.ids > $O/r06e_kstep_f6.txt; cat $O/r06e_kstep_f6.txt
timeout 200 python3 tools/trace_kstep.py --bit 8 --nout 41 --cfg wr128x192_t36_kstep_timeline,wr128x192_t_wave2 2>&1 | grep -v amdgpu.ids > $O/r06e_kstep_i8.txt; cat $O/r06e_kstep_i8.txt
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; bit=$2; shift 2; timeout 150 rocprofv3 --pmc "$@" -d /tmp/prof_$tag -o pmc -- python3 $R/tools/launch_few.py --bit $bit > $O/r06e_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*.db" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 $R/tools/rocprof_summary.py $f "gemm_wreg" 110 > $O/r06e_pmc_$tag.txt 2>&1; else tail -3 $O/r06e_$tag.log > $O/r06e_pmc_$tag.txt; fi
  rm -rf /tmp/prof_$tag; echo "== $tag"; grep -v "^#" $O/r06e_pmc_$tag.txt | head -16; }
prof sq4 4 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
prof lds4 4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
prof ta4 4 TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
prof sq8 8 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
cd $R
timeout 900 python3 -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r06e_suite.txt 2>&1; echo "suite rc=$?"; grep -v amdgpu.ids $O/r06e_suite.txt | tail -15
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06e_bench_line.json 2> $O/r06e_bench.err; echo "bench rc=$?"; tail -3 $O/r06e_bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06e_bench_line.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'gemm', d['roofline']['us_per_launch'], d['roofline']['frac'], 'single', d.get('tflops_single_replay'))
for k,v in (d.get('secondary_configs') or {}).items(): print(k, v['tflops'], v['ms_per_step'], v['gemm']['us_per_launch'], v['gemm']['frac'])
print('cold', d['timing']['cold_weights_ms_per_step'], 'mlp', d['timing']['mlp_block_ms'])
PY
