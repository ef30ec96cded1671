#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q > $O/r03d_pytest_r3.txt 2>&1; echo "rc=$?" >> $O/r03d_pytest_r3.txt
for c in "256x256x256 1" "256x256x4096 1" "512x1536x1024 1" "512x1536x4096 1" "512x1536x4096 5" "512x256x4096 1" "256x1536x4096 1" "1024x11008x4096 1" "2048x11008x4096 1" "4096x11008x4096 1"; do
  set -- $c
  timeout 120 python tools/dbg_fat.py wr256x256_s6_d3_self $1 $2 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/r03d_fat.txt; echo "   -> rc=$? ($c)" >> $O/r03d_fat.txt
done
tail -5 $O/r03d_pytest_r3.txt; cat $O/r03d_fat.txt
