set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_pair.py -x -q 2>&1 | tail -15
timeout 300 python tools/bench_pair.py 2>&1 | tail -6
timeout 300 python tools/bench_pair.py --amax 2>&1 | tail -6
