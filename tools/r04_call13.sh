#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p67_epi2_loaders_copy_2_of_4,wr128x192_p60_epi1,wr128x192_abl6_no_kloop_barriers,wr128x192_abl3_mfma > $O/r04l_ab.txt 2>&1
WR=$(MIXQ_TUNING_LIB=1 python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_s16_d4_l2','wr128x192_abl6_no_kloop_barriers')))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 --panels 2>&1 | grep -v "amdgpu.ids" > $O/r04l_trace.txt
cat $O/r04l_ab.txt $O/r04l_trace.txt
