#!/bin/bash
# after the tuple-ring FP6 form: full GPU suite, then the W4A4 evidence set again
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1400 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_final.txt 2>&1
sed -i 's/wr128x192_f6_s10,wr128x192_f6_d3,wr128x192_f6_l4/wr128x192_f6mov_d2,wr128x192_f6r_d3,wr128x192_f6r_s10,wr128x192_f6_l4/' tools/r03_call_f6.sh
bash tools/r03_call_f6.sh > gpurun_out/r03_final_b.log 2>&1
timeout 400 python tools/time_w4_small_batch.py > gpurun_out/r03_w4_small_batch.txt 2>&1
tail -4 gpurun_out/r03_gpu_final.txt; tail -12 gpurun_out/r03_final_b.log
