#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/r03i_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r03i_pytest.txt
tail -6 $O/r03i_pytest.txt
