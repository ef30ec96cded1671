#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/r03g_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r03g_pytest.txt
python bench.py > $O/r03g_bench.json 2> $O/r03g_bench.err
python tools/bench_configs.py > $O/r03g_configs.txt 2>&1
python tools/bench_mlp.py > $O/r03g_mlp.txt 2>&1
tail -4 $O/r03g_pytest.txt; tail -c 700 $O/r03g_bench.json; cat $O/r03g_configs.txt | tail -25; tail -8 $O/r03g_mlp.txt
