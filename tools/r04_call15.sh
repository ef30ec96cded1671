#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04m_pytest.txt 2>&1
tail -4 $O/r04m_pytest.txt
python bench.py > $O/r04m_bench.json 2> $O/r04m_bench.err
tail -c 1800 $O/r04m_bench.json
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1 > $O/r04m_ab.txt 2>&1
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1 --nout 0 >> $O/r04m_ab.txt 2>&1
cat $O/r04m_ab.txt
