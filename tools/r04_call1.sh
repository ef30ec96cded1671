#!/bin/bash
# Round-4 call 1: state of the kernel at the start of the round on this box + what the launch floor is made of.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 120 ./tools/ubench_launch > $O/r04a_launch.txt 2>&1
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_abl3_mfma,wr128x192_abl9_empty,wr128x192_abl7_nostore > $O/r04a_ab.txt 2>&1
WR=$(MIXQ_TUNING_LIB=1 python -c "
from mixq_amd import _capi
n=_capi.gemm_config_names()
print(','.join(str(i) for i,x in enumerate(n) if x in ('wr128x192_s16_d4_l2',)))")
python tools/trace_gemm.py --shapes 512x11008x4096 --cfgs $WR --nout 41 > $O/r04a_trace.txt 2>&1
python bench.py > $O/r04a_bench.json 2> $O/r04a_bench.err
cat $O/r04a_launch.txt $O/r04a_ab.txt $O/r04a_trace.txt; tail -c 1500 $O/r04a_bench.json
