#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; rm -f $O/r03m_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "int32_gemm_exact or fused_linear_vs_oracle or randomized_shapes or stream_k or prefill or ragged" > $O/r03m_pytest.txt 2>&1; echo "rc=$?" >> $O/r03m_pytest.txt
for shp in 4096x11008x4096 2048x11008x4096 8192x11008x4096 4096x4096x4096 2048x28672x8192; do
  timeout 900 python tools/ab_gemm.py --shape $shp --rounds 10 --launches 10 --cfgs wr128x256_s16_d3_l2,256x256_w4x2_s5_l0,256x128_w4x2_s5_l0,wr128x192_s16_d4_l2 >> $O/r03m_ab.txt 2>&1
done
tail -3 $O/r03m_pytest.txt; grep -v amdgpu $O/r03m_ab.txt
