#!/bin/bash
# The GPU suite with its output kept (gpurun_out/suite.txt); faulthandler shows the test a native crash happened in.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1200 python3 -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider "$@" > $O/suite.txt 2>&1
echo "rc=$?"; grep -v "amdgpu.ids" $O/suite.txt | tail -40
