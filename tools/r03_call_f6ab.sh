#!/bin/bash
export MIXQ_TUNING_LIB=1
mkdir -p gpurun_out
{
python tools/dbg_f6.py 512x11008x4096x128,200x328x512x143,512x4096x11008x128
for shp in 512x11008x4096 512x4096x11008; do
timeout 300 python tools/ab_gemm.py --shape $shp --bit 4 --f6 --nout 128 --rounds 20 --cfgs wr128x192_s16_d4_l2,wr128x192_f6_abl3_mfma,wr64x128_s16_d4_l2,128x192_w2x2_s5_l4
done
} > gpurun_out/r03_f6_ab2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03_f6_ab2.txt | tail -40
