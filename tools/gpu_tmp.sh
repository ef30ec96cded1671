cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "apply or operator or packed_only or checkpoint" -x 2>&1 | tail -3
timeout 600 python tools/bench_mlp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02aa_mlp.txt
timeout 300 python tools/time_norm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02aa_mlp.txt
