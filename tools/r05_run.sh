#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
bash tools/asan_capi.sh run
bash tools/gpu_suite.sh
