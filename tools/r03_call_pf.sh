#!/bin/bash
# L2-prefetch wave probe: correctness, cold-weights A/B, warm A/B
export MIXQ_TUNING_LIB=1
mkdir -p gpurun_out
{
for s in 512x11008x4096 512x4096x11008 500x4100x4096 2048x11008x4096; do timeout 120 python tools/dbg_fat.py wr128x192_p40_l2prefetch $s 3; done
echo "== cold (12 weight copies rotating)"
timeout 300 python tools/time_decode.py --shapes 512x11008x4096,512x4096x11008,512x4096x4096 --cfgs wr128x192_s16_d4_l2,wr128x192_p40_l2prefetch --copies 12 --nout 41 --rounds 20
echo "== warm"
timeout 300 python tools/ab_gemm.py --shape 512x11008x4096 --cfgs wr128x192_s16_d4_l2,wr128x192_p40_l2prefetch --rounds 30
} > gpurun_out/r03_pf.txt 2>&1
tail -40 gpurun_out/r03_pf.txt
