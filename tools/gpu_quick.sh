#!/bin/bash
# Quick GPU check after a kernel change: the quantise / norm parity tests, the driver's bench line three times, the quantise family's
# durations inside the real forwards (rocprofv3 --kernel-trace), and optionally the whole GPU suite.   usage: gpu_quick.sh TAG [suite]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-q}; cd $R
timeout 900 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -5 > $O/${T}_pytest.txt
cat $O/${T}_pytest.txt
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', $i, 'TFLOPS', d['value'], 'step', round(d['ms_per_step']*1e3,3), 'gemm', d['roofline']['us_per_launch'], 'min', round(d['timing']['replay_ms_min']*50,3))"; done > $O/${T}_bench.txt 2>&1
cat $O/${T}_bench.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpus 1 --no-cpu-baseline --no-secondary"
timeout 300 rocprofv3 --kernel-trace -d $O/prof_${T}_kt -o kt -- $B --steps 20 --warmup 5 > $O/${T}_kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/prof_${T}_kt2 -o kt -- $B --steps 20 --warmup 5 --shape 11008,4096 > $O/${T}_kt2.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/prof_${T}_kt3 -o kt -- python3 $R/tools/bench_mlp.py > $O/${T}_kt3.log 2>&1
cd $R
for n in kt kt2 kt3; do f=$(find gpurun_out/prof_${T}_$n -name "*.db" | head -1); python3 tools/rocprof_summary.py $f "" 110 > $O/${T}_$n.txt 2>&1; done
rm -rf gpurun_out/prof_${T}_*
grep -h "quant_\|rmsnorm\|gemm_wreg" $O/${T}_kt.txt $O/${T}_kt2.txt $O/${T}_kt3.txt | cut -c1-60,100-170
grep "norm + MLP" $O/${T}_kt3.log | cut -c1-200
if [ "$2" = suite ]; then timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/${T}_suite.txt; cat $O/${T}_suite.txt; fi
