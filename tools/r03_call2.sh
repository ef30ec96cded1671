#!/bin/bash
# Round-3 call 2: GPU tests on the new code (one-call forward, fused dequant/tail epilogue), the bench line with its new secondary
# timings, before/after A/B of the epilogue change between two builds, power / clock with the sampler on the right card.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03b_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r03b_pytest.txt
python tools/ab_libs.py --libs new=mixq_amd/libmixq_hip.so,r02=mixq_amd/libmixq_prev_r02epi.so --shapes 512x11008x4096 --nouts 0,41 > $O/r03b_ab_epilogue.txt 2>&1
python tools/ab_libs.py --libs new=mixq_amd/libmixq_hip.so,r02=mixq_amd/libmixq_prev_r02epi.so --shapes 512x4096x4096,512x4096x11008,512x28672x8192 --nouts 41,110 --rounds 15 >> $O/r03b_ab_epilogue.txt 2>&1
python bench.py > $O/r03b_bench.json 2> $O/r03b_bench.err
python tools/yardstick.py --shapes 512x11008x4096,4096x11008x4096 --rounds 5 > $O/r03b_power.txt 2>&1
tail -5 $O/r03b_pytest.txt; cat $O/r03b_ab_epilogue.txt; tail -c 900 $O/r03b_bench.json; grep -E "sampler|sustained|TOPS" $O/r03b_power.txt | tail -20
