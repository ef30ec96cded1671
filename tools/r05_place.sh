#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for r in 1 2 3; do for lib in libmixq_hip_kpq.so libmixq_hip.so; do
  for extra in "" "--shape 4096,4096" "--shape 11008,4096"; do
  MIXQ_LIB_FILE=$lib python3 bench.py --no-cpu-baseline --no-secondary $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', $r, '$extra', 'step', round(d['ms_per_step']*1e3,3), 'gemm', d['roofline']['us_per_launch'], 'min', round(d['timing']['replay_ms_min']*50,3))"
done; done; done > $O/r05j_place.txt 2>&1
cat $O/r05j_place.txt
python3 tools/bench_mlp.py 2>&1 | grep -v amdgpu.ids | tail -6
