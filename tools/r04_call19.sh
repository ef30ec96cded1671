#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q > $O/r04p_pytest4.txt 2>&1
tail -30 $O/r04p_pytest4.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04p_pytest.txt 2>&1
tail -5 $O/r04p_pytest.txt
