#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_fp6.py -m gpu -q -x 2>&1 | tail -4
export MIXQ_TUNING_LIB=1
for shp in 512x4096x11008 512x4096x4096; do
timeout 300 python tools/ab_gemm.py --shape $shp --bit 4 --f6 --nout 128 --rounds 20 --cfgs wr64x128_s16_d4_l2,wr64x128_f6_d2,wr64x128_f6_d3,wr64x128_f6_d5,64x128_w2x2_s5_l4
done
} > gpurun_out/r03_f6_ab4.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03_f6_ab4.txt | tail -40
