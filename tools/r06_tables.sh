#!/bin/bash
# Round-6 measurement tables (one box): operator table over the BASELINE + published shapes, the reference's batch sweep at operator level,
# the Option-A (unchanged reference call sequence) route, the vendor yardstick, MLP block, decoder-layer path.  -> gpurun_out/r06u_*
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python3 tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > $O/r06u_configs_operator.txt
python3 tools/bench_configs.py --msweep 2>&1 | grep -v amdgpu.ids > $O/r06u_msweep_operator.txt
python3 tools/bench_configs.py --M 2048 --only "published" 2>&1 | grep -v amdgpu.ids > $O/r06u_published_m2048.txt
python3 tools/time_arch9.py 2>&1 | grep -v amdgpu.ids > $O/r06u_arch9_route.txt
python3 tools/yardstick.py --no-power --rounds 10 2>&1 | grep -v amdgpu.ids > $O/r06u_ceiling.txt
python3 tools/bench_mlp.py 2>&1 | grep -v amdgpu.ids > $O/r06u_mlp_block.txt
python3 tools/bench_block.py --layers 32 --passes 1 2>&1 | grep -v amdgpu.ids > $O/r06u_block.txt
python3 tools/prefill_sweep.py 2>&1 | grep -v amdgpu.ids > $O/r06u_prefill_sweep.txt
tail -5 $O/r06u_configs_operator.txt; tail -3 $O/r06u_msweep_operator.txt; tail -3 $O/r06u_arch9_route.txt; tail -8 $O/r06u_ceiling.txt
