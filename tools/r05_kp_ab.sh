#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for r in 1 2 3; do for lib in libmixq_hip.so libmixq_hip_kp.so; do
  echo "== $lib (round $r)"; MIXQ_LIB_FILE=$lib python3 tools/time_quant_mask.py --rounds 5 2>&1 | grep -E "median"
  MIXQ_LIB_FILE=$lib python3 tools/time_quant_mask.py --rounds 5 --shape 4096,4096 2>&1 | grep -E "median"
done; done > $O/r05c_kp_ab.txt 2>&1
cat $O/r05c_kp_ab.txt
