#!/bin/bash
# Round-3 evidence run on the GPU box (one box, one call): everything lands in gpurun_out/r03p_*; copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_r03p_kt -o kt -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $O/r03p_kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/prof_r03p_fetch -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $O/r03p_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/prof_r03p_write -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $O/r03p_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE -d $O/prof_r03p_mfma -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $O/r03p_mfma.log 2>&1
timeout 400 rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/prof_r03p_mem -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $O/r03p_mem.log 2>&1
cd $R
for n in kt fetch write mfma mem; do f=$(find gpurun_out/prof_r03p_$n -name "*.db" | head -1); python tools/rocprof_summary.py $f > gpurun_out/r03p_$n.txt 2>&1; done
rm -rf gpurun_out/prof_r03p_*
python tools/make_traffic_json.py gpurun_out/r03p_fetch.txt gpurun_out/r03p_write.txt gpurun_out/r03p_hbm_traffic.json "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over python bench.py --steps 20 --warmup 2 --no-graph (tools/profile_bench_r03.sh); summaries in profiles/r03_bench_pmc_fetch.txt, r03_bench_pmc_write.txt" > gpurun_out/r03p_traffic.log 2>&1
python bench.py > gpurun_out/r03p_bench.json 2> gpurun_out/r03p_bench.err
python bench.py --bit 4 --no-cpu-baseline > gpurun_out/r03p_bench_w4.json 2>> gpurun_out/r03p_bench.err
python bench.py --shape 8192,28672 --no-cpu-baseline --steps 100 > gpurun_out/r03p_bench_70b.json 2>> gpurun_out/r03p_bench.err
python tools/bench_configs.py > gpurun_out/r03p_configs.txt 2>&1
python tools/ab_gemm.py --cfgs 128x192_w2x2_s5_l4,wr128x192_s16_d4_l2,wr128x192_abl1_noW,wr128x192_abl2_noX,wr128x192_abl3_mfma,wr128x192_abl9_empty > gpurun_out/r03p_ab.txt 2>&1
WR=$(MIXQ_TUNING_LIB=1 python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_s16_d4_l2','wr128x192_abl3_mfma')))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR > gpurun_out/r03p_trace.txt 2>&1
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 >> gpurun_out/r03p_trace.txt 2>&1
python tools/time_quant.py > gpurun_out/r03p_quant.txt 2>&1
head -8 gpurun_out/r03p_kt.txt; cat gpurun_out/r03p_traffic.log | tail -12; tail -c 1500 gpurun_out/r03p_bench.json
