#!/bin/bash
# GPU session B of round 2
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/r02b_pytest.txt 2>&1
echo "rc=$?" >> $O/r02b_pytest.txt
timeout 300 python tools/time_quant.py > $O/r02b_quant.txt 2>&1
timeout 900 python tools/sweep_gemm.py --shapes 512x11008x4096 --packed 1,2 --nout 41 --out $O/r02b_sweep.json > $O/r02b_sweep.txt 2>&1
timeout 1500 python tools/sweep_gemm.py --shapes 512x4096x4096,512x12288x4096,512x4096x11008,512x6144x4096,512x14336x4096,512x4096x14336,512x8192x8192,512x28672x8192,512x10240x8192,512x8192x28672 --packed 1,2 --out $O/r02b_sweep_shapes.json > $O/r02b_sweep_shapes.txt 2>&1
timeout 600 python bench.py > $O/r02b_bench.json 2> $O/r02b_bench.err
tail -5 $O/r02b_pytest.txt; tail -c 300 $O/r02b_bench.json
