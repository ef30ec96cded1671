#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for r in 1 2; do for lib in libmixq_hip_kpq.so libmixq_hip.so; do
  rm -rf $O/prof_kp
  MIXQ_LIB_FILE=$lib timeout 300 rocprofv3 --kernel-trace -d $O/prof_kp -o kt -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $O/r05m_$lib.$r.log 2>&1
  f=$(find $O/prof_kp -name "*.db" | head -1)
  echo "== $lib round $r"; python3 $R/tools/rocprof_summary.py $f | grep -E "quant_rows2|gemm_wreg" | head -3
  python3 $R/tools/driver_gaps.py $f 20 | grep -E "back to back|all replays"
done; done > $O/r05m_kp_trace.txt 2>&1
rm -rf $O/prof_kp
cat $O/r05m_kp_trace.txt
