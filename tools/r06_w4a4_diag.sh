#!/bin/bash
# Round-6 diagnosis of the W4A4 (FP6-pipe) loop, one box, one call: the tuple-ring form's feed ablations and ring
# variants interleaved (tuning library), the in-k-step timeline, the phase trace, and SQ / TA / TCP counter passes of the SHIPPED kernel
# (product library, separate --pmc runs).  Output: gpurun_out/r06d_*.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
C=wr128x192_s16_d4_l2,wr128x192_f6r_abl3_mfma,wr128x192_f6r_abl1_noW,wr128x192_f6r_abl2_noX,wr128x192_f6r_abl32_dma_noreads,wr128x192_f6r_abl33_reads_nodma,wr128x192_f6r_abl6_nobar,wr128x192_f6r_xr8,wr128x192_f6r_xr2,wr128x192_f6r_l4,wr128x192_f6r_d3,wr128x192_f6r_d4,wr128x192_f6r_s6,wr128x192_f6r_s10,wr128x192_f6r_t36_kstep_timeline
MIXQ_TUNING_LIB=1 timeout 600 python3 tools/prefill_sweep.py --bit 4 --tokens 512 --layers 11008x4096 --cfgs $C --rounds 15 2>&1 | grep -v amdgpu.ids > $O/r06d_f6_variants.txt
cat $O/r06d_f6_variants.txt | tr -s ' ' | sed 's/  /\n/g' | head -60
timeout 300 python3 tools/trace_kstep.py 2>&1 | grep -v amdgpu.ids > $O/r06d_kstep.txt; cat $O/r06d_kstep.txt
timeout 300 python3 tools/trace_kstep.py --nout 0 2>&1 | grep -v amdgpu.ids >> $O/r06d_kstep.txt
timeout 300 python3 tools/trace_gemm.py --f6 --bit 4 --nout 128 --shapes 512x11008x4096 --cfgs wr128x192_s16_d4_l2,wr128x192_f6r_abl3_mfma 2>&1 | grep -v amdgpu.ids > $O/r06d_gemm_trace.txt; cat $O/r06d_gemm_trace.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/r06d_avail.txt 2>&1
B="python3 $R/bench.py --gpus 1 --no-cpu-baseline --no-secondary --bit 4 --steps 20 --warmup 2 --no-graph"
prof() { tag=$1; shift; timeout 400 rocprofv3 --pmc "$@" -d $O/prof_r06d_$tag -o pmc -- $B > $O/r06d_$tag.log 2>&1; }
prof sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
prof inst SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
prof lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS
prof mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM
prof ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
prof tcp TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
prof tcp2 TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
prof td TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum
cd $R
for n in sq inst lds mfma ta tcp tcp2 td; do
    f=$(find gpurun_out/prof_r06d_$n -name "*.db" 2>/dev/null | head -1)
    if [ -n "$f" ]; then python3 tools/rocprof_summary.py $f "gemm_wreg" 110 > $O/r06d_pmc_$n.txt 2>&1; else tail -5 $O/r06d_$n.log > $O/r06d_pmc_$n.txt; fi
done
rm -rf gpurun_out/prof_r06d_*
for n in sq inst lds mfma ta tcp tcp2 td; do echo "== $n"; grep -v "^#" $O/r06d_pmc_$n.txt | head -24; done
