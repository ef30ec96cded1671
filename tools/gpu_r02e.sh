#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python tools/time_quant.py --cold > $O/r02e_quant_cold.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_r02e_kt -o kt -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/$O/r02e_kt.log 2>&1
cd $R
f=$(find $O/prof_r02e_kt -name "*.db" | head -1); python tools/rocprof_summary.py $f > $O/r02e_kt.txt 2>&1
cat $O/r02e_quant_cold.txt | cut -c1-250; head -12 $O/r02e_kt.txt; tail -c 700 $O/r02e_kt.log
