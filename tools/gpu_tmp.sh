cd $GRAFT_REPO_ROOT
timeout 1200 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02w_configs.txt
timeout 600 python tools/time_arch9.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02w_arch9.txt
