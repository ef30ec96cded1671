#!/bin/bash
# Round-4 call 3: EPI2 probes (tuning build): where the panel pipeline loses its time.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1,wr128x192_p61_epi2_copy_at_end,wr128x192_p62_epi2_loaders_prio0 > $O/r04c_ab.txt 2>&1
WR=$(MIXQ_TUNING_LIB=1 python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_p61_epi2_copy_at_end','wr128x192_p62_epi2_loaders_prio0')))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 > $O/r04c_trace.txt 2>&1
cat $O/r04c_ab.txt $O/r04c_trace.txt
