#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/r03s_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r03s_pytest.txt
python tools/time_secondary.py 2>&1 | grep -v amdgpu > $O/r03s_secondary.txt
tail -4 $O/r03s_pytest.txt; cat $O/r03s_secondary.txt
