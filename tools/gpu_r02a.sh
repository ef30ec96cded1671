#!/bin/bash
# GPU session A of round 2: capability, new-kernel smoke, full gpu test suite, quantise / GEMM sweeps, arch9 timing, bench line.
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -c "import torch;print(torch.cuda.get_device_capability(), torch.cuda.get_device_name())" > $O/r02a_cap.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -k "wreg or packed_quantise or fused_linear_vs_oracle" > $O/r02a_pytest_new.txt 2>&1
echo "rc=$?" >> $O/r02a_pytest_new.txt
timeout 300 python tools/time_quant.py > $O/r02a_quant.txt 2>&1
timeout 900 python tools/sweep_gemm.py --shapes 512x11008x4096 --packed 1,2 --nout 41 --out $O/r02a_sweep.json > $O/r02a_sweep.txt 2>&1
timeout 300 python tools/time_arch9.py > $O/r02a_arch9.txt 2>&1
timeout 600 python bench.py > $O/r02a_bench.json 2> $O/r02a_bench.err
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/r02a_pytest.txt 2>&1
echo "rc=$?" >> $O/r02a_pytest.txt
tail -3 $O/r02a_pytest_new.txt; tail -5 $O/r02a_pytest.txt; tail -c 400 $O/r02a_bench.json
