#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04e_pytest.txt 2>&1
tail -5 $O/r04e_pytest.txt
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1 > $O/r04e_ab.txt 2>&1
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1 --nout 0 >> $O/r04e_ab.txt 2>&1
cat $O/r04e_ab.txt
