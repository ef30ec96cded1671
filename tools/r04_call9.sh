#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04i_pytest.txt 2>&1
tail -4 $O/r04i_pytest.txt
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1,wr128x192_p61_epi2_copy_at_end > $O/r04i_ab.txt 2>&1
WR=$(MIXQ_TUNING_LIB=1 python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_s16_d4_l2',)))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 > $O/r04i_trace.txt 2>&1
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 --panels >> $O/r04i_trace.txt 2>&1
cat $O/r04i_ab.txt $O/r04i_trace.txt
