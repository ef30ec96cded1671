#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
{
python tools/ab_gemm.py --shape 4096x11008x4096 --cfgs 256x256_w4x2_s5_l0 --rounds 10
python tools/ab_gemm.py --shape 4096x8192x4096 --cfgs 256x256_w4x2_s5_l0 --rounds 10
python tools/ab_gemm.py --shape 4096x2816x4096 --cfgs sk256x256_w4x2_s4,256x256_w4x2_s5_l0,sk256x128_w4x2_s5 --rounds 10
python tools/ab_gemm.py --shape 8192x11008x4096 --cfgs 256x256_w4x2_s5_l0 --rounds 10
python tools/ab_gemm.py --shape 8192x10240x4096 --cfgs 256x256_w4x2_s5_l0 --rounds 10
python tools/ab_gemm.py --shape 8192x768x4096 --cfgs sk256x256_w4x2_s4,256x256_w4x2_s5_l0,sk256x128_w4x2_s5 --rounds 10
} 2>&1 | grep -v amdgpu > $O/r04u_sk_probe.txt
cat $O/r04u_sk_probe.txt
