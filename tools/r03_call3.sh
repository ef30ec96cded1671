#!/bin/bash
# Round-3 call 3: the prefill tile (256 x 256, four fat self-loading waves) and the grouped tile order: correctness, then interleaved A/B.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -k "wreg or wr_ or randomized or more_than_128 or packed_only or round3 or one_call or compacted or remaining or prefill or ragged" > $O/r03c_pytest_sel.txt 2>&1; echo "rc=$?" >> $O/r03c_pytest_sel.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/r03c_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r03c_pytest.txt
for shp in 4096x11008x4096 2048x11008x4096; do
  timeout 600 python tools/ab_gemm.py --shape $shp --rounds 12 --launches 10 --gms 32,8 --cfgs wr128x256_s16_d3_l2 >> $O/r03c_ab_prefill.txt 2>&1
  timeout 600 python tools/ab_gemm.py --shape $shp --rounds 12 --launches 10 --gms 0,4,16 --cfgs wr128x256_s16_d3_l2,wr256x256_s6_d3_self,wr256x256_s7_d4_self,256x256_w4x2_s5_l0 >> $O/r03c_ab_prefill.txt 2>&1
done
timeout 600 python tools/ab_gemm.py --shape 512x28672x8192 --rounds 12 --launches 10 --nout 82 --cfgs wr128x256_s16_d3_l2,wr256x256_s6_d3_self,wr256x256_s7_d4_self >> $O/r03c_ab_prefill.txt 2>&1
timeout 600 python tools/ab_gemm.py --shape 1024x11008x4096 --rounds 12 --launches 10 --gms 0 --cfgs wr128x192_s16_d4_l2,wr128x256_s16_d3_l2,wr256x256_s6_d3_self >> $O/r03c_ab_prefill.txt 2>&1
tail -4 $O/r03c_pytest_sel.txt; tail -4 $O/r03c_pytest.txt; cat $O/r03c_ab_prefill.txt
