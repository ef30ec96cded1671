#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04s_pytest.txt 2>&1
tail -4 $O/r04s_pytest.txt
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1 > $O/r04s_ab.txt 2>&1
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1 --nout 0 >> $O/r04s_ab.txt 2>&1
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,wr128x192_p60_epi1 --nout 20 >> $O/r04s_ab.txt 2>&1
grep -v amdgpu $O/r04s_ab.txt
python bench.py --no-cpu-baseline > $O/r04s_bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/r04s_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['us_per_launch'])"
