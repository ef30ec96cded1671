#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
WR=$(python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x.startswith('wr128x192')))")
timeout 600 python tools/sweep_gemm.py --shapes 512x11008x4096 --cfgs $WR --packed 2 --nout 41 --krot 0,1,3,8,11,16,21,32 --out $O/r02d_sweep.json > $O/r02d_sweep.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "wreg or every_k or randomized_shapes or more_than_128" > $O/r02d_pytest.txt 2>&1
cat $O/r02d_sweep.txt | cut -c1-170; tail -3 $O/r02d_pytest.txt
