#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 600 python tools/bench_block.py > $O/r04q_block.txt 2>&1
cat $O/r04q_block.txt | grep -v amdgpu.ids
timeout 300 python tools/bench_mlp.py > $O/r04q_mlp.txt 2>&1
cat $O/r04q_mlp.txt | grep -v amdgpu.ids
timeout 600 python tools/bench_configs.py > $O/r04q_configs.txt 2>&1
cat $O/r04q_configs.txt | grep -v amdgpu.ids
