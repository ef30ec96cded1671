cd $GRAFT_REPO_ROOT
timeout 300 python tools/ab_gemm.py --shape 512x11008x4096 --cfgs 128x192_w2x2_s5_l4,wr128x192_s16_d4_l2,wr128x192_abl10_drain1,wr128x192_abl11_drain2 --rounds 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02s_ab2.txt
