#!/bin/bash
# Round-5 first evidence run: the driver's exact invocation, un-profiled and under rocprofv3 --kernel-trace; GPU suite.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05a_bench_line.json 2> $O/r05a_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/r05a_bench_line2.json 2>> $O/r05a_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/prof_r05a_kt -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/r05a_kt.log 2>&1
cd $R
f=$(find gpurun_out/prof_r05a_kt -name "*.db" | head -1)
python3 tools/driver_gaps.py $f 20 > $O/r05a_driver_gaps.txt 2>&1
python3 tools/rocprof_summary.py $f > $O/r05a_kt.txt 2>&1
rm -rf gpurun_out/prof_r05a_kt
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/r05a_pytest.txt 2>&1
tail -3 $O/r05a_pytest.txt
cat $O/r05a_driver_gaps.txt
python3 -c "
import json
for f in ('r05a_bench_line.json','r05a_bench_line2.json'):
    d=json.loads(open('$O/'+f).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v for k,v in d['timing'].items() if 'ms' in k and 'protocol' not in k}, d['roofline']['us_per_launch'])
"
