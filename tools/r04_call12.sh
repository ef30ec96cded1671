#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1,wr128x192_p64_epi2_no_tail_mfma,wr128x192_p65_epi2_no_staging_writes,wr128x192_p66_epi2_no_scaling,wr128x192_abl7_nostore > $O/r04k_ab.txt 2>&1
WR=$(MIXQ_TUNING_LIB=1 python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_s16_d4_l2','wr128x192_p64_epi2_no_tail_mfma','wr128x192_p65_epi2_no_staging_writes','wr128x192_p66_epi2_no_scaling')))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 2>&1 | grep -v "amdgpu.ids" | grep "cfg\|dequant\|loop end\|store issue\|WG total" > $O/r04k_trace.txt
cat $O/r04k_ab.txt $O/r04k_trace.txt
