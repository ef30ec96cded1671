#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q > $O/r03h_pytest_r3.txt 2>&1; echo "rc=$?" >> $O/r03h_pytest_r3.txt
python tools/bench_mlp.py > $O/r03h_mlp.txt 2>&1
python tools/bench_mlp.py >> $O/r03h_mlp.txt 2>&1
tail -15 $O/r03h_pytest_r3.txt; grep -v amdgpu $O/r03h_mlp.txt
