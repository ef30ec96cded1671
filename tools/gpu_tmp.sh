cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "prefill or apply" -x 2>&1 | tail -12
