#!/bin/bash
# Round-3 final evidence (one box, one call): the headline profile set of tools/profile_bench_r03.sh, then the FP6-pipe W4A4 set.
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_bench_r03.sh > $R/gpurun_out/r03_final_a.log 2>&1
bash $R/tools/r03_call_f6.sh > $R/gpurun_out/r03_final_b.log 2>&1
cd $R
timeout 100 ./tools/ubench_fp6 > gpurun_out/r03_ubench_fp6.txt 2>&1
timeout 400 python tools/time_w4_small_batch.py > gpurun_out/r03_w4_small_batch.txt 2>&1
timeout 300 python tools/bench_mlp.py > gpurun_out/r03p_mlp.txt 2>&1
tail -5 gpurun_out/r03_final_a.log; tail -12 gpurun_out/r03_final_b.log
