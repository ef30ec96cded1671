#!/bin/bash
# Round-3 call 1: yardsticks (vendor int8 GEMM, bare MFMA, power / clock) + the state of the bench line at the start of the round.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python tools/yardstick.py > $O/r03a_yardstick.txt 2>&1
python bench.py > $O/r03a_bench.json 2> $O/r03a_bench.err
# the vendor kernel's name (hipBLASLt encodes its macro-tile, MFMA shape and schedule in it)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_r03a_kt -o kt -- python $R/tools/yardstick.py --shapes 512x11008x4096,4096x11008x4096 --rounds 2 --launches 5 --no-power > $O/r03a_kt.log 2>&1
cd $R
f=$(find gpurun_out/prof_r03a_kt -name "*.db" | head -1); python tools/rocprof_summary.py $f "" 400 > gpurun_out/r03a_kt.txt 2>&1
rm -rf gpurun_out/prof_r03a_kt
tail -40 $O/r03a_yardstick.txt; tail -c 600 $O/r03a_bench.json
