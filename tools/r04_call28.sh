#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_s16_d4_l4,wr128x192_s12_d4_l2,wr128x192_s16_d3_l2,wr128x192_p60_epi1 2>&1 | grep -v amdgpu > $O/r04t_ab.txt
cat $O/r04t_ab.txt
