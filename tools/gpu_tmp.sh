cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "packed_operands or arch9 or reference_style" 2>&1 | tail -15
python tools/time_arch9.py 2>&1 | grep -v amdgpu.ids
