#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/r03n_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r03n_pytest.txt
python tools/yardstick.py --shapes 4096x11008x4096,8192x11008x4096,4096x4096x4096,2048x11008x4096 --rounds 10 --no-power > $O/r03n_yard.txt 2>&1
tail -5 $O/r03n_pytest.txt; grep -v amdgpu $O/r03n_yard.txt
