#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "quant or extract or rmsnorm or operator or trace or wreg" > $O/r02f_pytest.txt 2>&1
timeout 300 python tools/time_quant.py > $O/r02f_quant.txt 2>&1
WR=$(python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_s8_d4_l1','wr128x192_s8_d4_l2','wr64x128_s8_d4_l1')))")
timeout 300 python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR > $O/r02f_trace.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/r02f_bench.json 2> $O/r02f_bench.err
tail -3 $O/r02f_pytest.txt; grep "fmt=1\|fmt=0" $O/r02f_quant.txt | cut -c1-220; cat $O/r02f_trace.txt | head -40; tail -c 1500 $O/r02f_bench.json
