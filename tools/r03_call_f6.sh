#!/bin/bash
# FP6-pipe W4A4 evidence: bench line, kernel trace of the same command, operator table, in-kernel timeline, feed ablations, quantiser formats
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_r03w4_kt -o kt -- python $R/bench.py --bit 4 --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $O/r03w4_kt.log 2>&1
cd $R
f=$(find gpurun_out/prof_r03w4_kt -name "*.db" | head -1); python tools/rocprof_summary.py $f > gpurun_out/r03w4_kt.txt 2>&1
rm -rf gpurun_out/prof_r03w4_kt
python bench.py --bit 4 --no-cpu-baseline > gpurun_out/r03w4_bench.json 2> gpurun_out/r03w4_bench.err
python tools/bench_configs.py > gpurun_out/r03w4_configs.txt 2>&1
export MIXQ_TUNING_LIB=1
C=$(python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_s16_d4_l2','wr128x192_f6_abl3_mfma')))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $C --bit 4 --f6 --nout 128 > gpurun_out/r03w4_trace.txt 2>&1
for shp in 512x11008x4096 512x4096x11008; do
python tools/ab_gemm.py --shape $shp --bit 4 --f6 --nout 128 --rounds 20 --cfgs wr128x192_s16_d4_l2,wr128x192_f6_abl1_noW,wr128x192_f6_abl2_noX,wr128x192_f6_abl3_mfma,wr128x192_f6mov_d2,wr128x192_f6r_d3,wr128x192_f6r_s10,wr128x192_f6_l4,wr64x128_s16_d4_l2,128x192_w2x2_s5_l4
done > gpurun_out/r03w4_ab.txt 2>&1
python tools/time_quant4.py > gpurun_out/r03w4_quant.txt 2>&1
head -6 gpurun_out/r03w4_kt.txt; cat gpurun_out/r03w4_bench.json | head -c 600; echo; grep -A4 "cfg2" gpurun_out/r03w4_configs.txt
