#!/bin/bash
# The GPU suite with its output kept (gpurun_out/suite.txt); faulthandler shows the test a native crash happened in.  Second pass: the
# kernels that live in the tuning build only (stream-K, pairwise split-K) against the same tests, with libmixq_hip_tuning.so loaded.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1200 python3 -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider "$@" > $O/suite.txt 2>&1
echo "rc=$?"; grep -v "amdgpu.ids" $O/suite.txt | tail -40
if [ -f mixq_amd/libmixq_hip_tuning.so ]; then
  MIXQ_TUNING_LIB=1 timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider -k "stream_k or split_k" > $O/suite_tuning.txt 2>&1
  echo "tuning-build pass rc=$?"; grep -v "amdgpu.ids" $O/suite_tuning.txt | tail -5
fi
