#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; rm -f $O/r03f_ab_prefill.txt
for shp in 4096x11008x4096 2048x11008x4096 8192x11008x4096; do
  timeout 900 python tools/ab_gemm.py --shape $shp --rounds 12 --launches 10 --gms 0,4,16 --cfgs wr128x256_s16_d3_l2,wr256x256_s6_d3_self,256x256_w4x2_s5_l0 >> $O/r03f_ab_prefill.txt 2>&1
done
timeout 600 python tools/ab_gemm.py --shape 4096x4096x4096 --rounds 12 --launches 10 --cfgs wr128x256_s16_d3_l2,wr128x128_s16_d4_l2,wr256x256_s6_d3_self >> $O/r03f_ab_prefill.txt 2>&1
timeout 600 python tools/ab_gemm.py --shape 512x28672x8192 --rounds 12 --launches 10 --nout 82 --cfgs wr128x256_s16_d3_l2,wr256x256_s6_d3_self >> $O/r03f_ab_prefill.txt 2>&1
timeout 600 python tools/ab_gemm.py --shape 4096x11008x4096 --rounds 8 --launches 10 --nout 0 --cfgs wr128x256_s16_d3_l2,wr256x256_s6_d3_self >> $O/r03f_ab_prefill.txt 2>&1
python tools/yardstick.py --shapes 4096x11008x4096 --rounds 8 --no-power >> $O/r03f_ab_prefill.txt 2>&1
grep -v amdgpu.ids $O/r03f_ab_prefill.txt
