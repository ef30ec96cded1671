#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; rm -f $O/r03e_fat.txt
for c in "512x1536x4096 5" "1024x11008x4096 1" "1024x11008x4096 3" "2048x11008x4096 1" "4096x11008x4096 1" "4096x11008x4096 3"; do
  set -- $c
  timeout 120 python tools/dbg_fat.py wr256x256_s6_d3_self $1 $2 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/r03e_fat.txt; echo "   -> rc=$? ($c)" >> $O/r03e_fat.txt
  timeout 120 python tools/dbg_fat.py wr256x256_s7_d4_self $1 $2 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/r03e_fat.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wreg or randomized_shapes or more_than_128" > $O/r03e_pytest_sel.txt 2>&1; echo "rc=$?" >> $O/r03e_pytest_sel.txt
cat $O/r03e_fat.txt; tail -5 $O/r03e_pytest_sel.txt
