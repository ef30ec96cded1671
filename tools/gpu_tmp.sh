cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "quantise or operator or golden or forward or baseline" -x 2>&1 | tail -3
timeout 300 python tools/ab_quant.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02z_ab_quant.txt
python bench.py 2>gpurun_out/r02z_bench.err | tee gpurun_out/r02z_bench.json | cut -c1-330
