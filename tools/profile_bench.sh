#!/bin/bash
# Round-2 evidence run on the GPU box: everything lands in gpurun_out/r02p_*; the summaries worth keeping are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace of the bench command (per-kernel average durations)
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_r02p_kt -o kt -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/r02p_kt.log 2>&1
# 2. HBM traffic of the dominant kernel: separate FETCH / WRITE passes (TCC slots), eager launches so every dispatch is counted
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/prof_r02p_fetch -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph --no-cpu-baseline > $O/r02p_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/prof_r02p_write -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph --no-cpu-baseline > $O/r02p_write.log 2>&1
# 3. MFMA utilisation counters of the same kernel
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE -d $O/prof_r02p_mfma -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph --no-cpu-baseline > $O/r02p_mfma.log 2>&1
timeout 400 rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/prof_r02p_mem -o pmc -- python $R/bench.py --steps 20 --warmup 2 --no-graph --no-cpu-baseline > $O/r02p_mem.log 2>&1
cd $R
for n in kt fetch write mfma mem; do f=$(find gpurun_out/prof_r02p_$n -name "*.db" | head -1); python tools/rocprof_summary.py $f > gpurun_out/r02p_$n.txt 2>&1; done
rm -rf gpurun_out/prof_r02p_*            # the databases are tens of MB each; only the summaries travel back
# 4. the bench line itself, per-config operator table, the reference's arch == 9 route, quantise geometries, interleaved GEMM A/B
python bench.py > gpurun_out/r02p_bench.json 2> gpurun_out/r02p_bench.err
python tools/bench_configs.py > gpurun_out/r02p_configs.txt 2>&1
python tools/time_arch9.py > gpurun_out/r02p_arch9.txt 2>&1
python tools/time_quant.py > gpurun_out/r02p_quant.txt 2>&1
python tools/time_quant.py --probe >> gpurun_out/r02p_quant.txt 2>&1
python tools/ab_gemm.py --cfgs 128x192_w2x2_s5_l4,wr128x192_s16_d4_l2,wr128x192_s8_d4_l1,wr128x192_abl1_noW,wr128x192_abl2_noX,wr128x192_abl3_mfma > gpurun_out/r02p_ab.txt 2>&1
python tools/ab_gemm.py --shape 512x4096x4096 --cfgs 64x128_w2x2_s5_l4,wr64x128_s16_d4_l2 >> gpurun_out/r02p_ab.txt 2>&1
python tools/ab_gemm.py --shape 512x4096x11008 --cfgs 64x128_w2x2_s5_l4,wr64x128_s16_d4_l2 --nout 110 >> gpurun_out/r02p_ab.txt 2>&1
python tools/ab_gemm.py --shape 512x8192x8192 --cfgs 128x128_w2x2_s5_l2,wr128x128_s16_d4_l2 --nout 82 >> gpurun_out/r02p_ab.txt 2>&1
python tools/ab_gemm.py --shape 512x28672x8192 --cfgs 256x256_w4x2_s5_l0,wr128x256_s16_d3_l2 --nout 82 --rounds 15 >> gpurun_out/r02p_ab.txt 2>&1
python tools/ab_gemm.py --shape 512x8192x28672 --cfgs 128x128_w2x2_s5_l2,wr128x128_s16_d4_l2 --nout 287 --rounds 15 >> gpurun_out/r02p_ab.txt 2>&1
WR=$(python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('128x192_w2x2_s5_l4','wr128x192_s16_d4_l2','wr128x192_abl1_noW','wr128x192_abl2_noX','wr128x192_abl3_mfma')))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR > gpurun_out/r02p_trace.txt 2>&1
head -8 gpurun_out/r02p_kt.txt; grep -A3 "gemm_wreg" gpurun_out/r02p_fetch.txt | tail -4; grep -A3 "gemm_wreg" gpurun_out/r02p_write.txt | tail -4; tail -c 300 gpurun_out/r02p_bench.json
