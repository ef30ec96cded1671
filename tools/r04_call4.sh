#!/bin/bash
# Round-4 call 4: EPI2 with the address-free copy-out and the loaders' X_out fix-up: parity suite + A/B + timeline.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04d_pytest.txt 2>&1
tail -5 $O/r04d_pytest.txt
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1,wr128x192_p61_epi2_copy_at_end,wr128x192_p62_epi2_loaders_prio0 > $O/r04d_ab.txt 2>&1
WR=$(MIXQ_TUNING_LIB=1 python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_s16_d4_l2', 'wr128x192_p61_epi2_copy_at_end')))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 > $O/r04d_trace.txt 2>&1
cat $O/r04d_ab.txt $O/r04d_trace.txt
