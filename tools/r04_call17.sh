#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
C=wr128x192_s16_d4_l2,wr128x192_s16_d3_l2,wr128x192_s16_d5_l2,wr128x192_s16_d6_l2,wr128x192_s12_d6_l2,wr128x192_p60_epi1
python tools/ab_gemm.py --cfgs $C --launches 24 > $O/r04n_ab_cold.txt 2>&1
python tools/ab_gemm.py --cfgs $C --launches 24 --cold 8 >> $O/r04n_ab_cold.txt 2>&1
grep -v amdgpu.ids $O/r04n_ab_cold.txt
