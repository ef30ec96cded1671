#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
WR=$(MIXQ_TUNING_LIB=1 python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_s16_d4_l2',)))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 2>&1 | grep -v "amdgpu.ids" > $O/r04o_trace.txt
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 --cold 8 2>&1 | grep -v "amdgpu.ids" >> $O/r04o_trace.txt
cat $O/r04o_trace.txt
