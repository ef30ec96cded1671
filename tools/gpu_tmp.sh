cd $GRAFT_REPO_ROOT
python tools/ab_gemm.py --shape 512x11008x4096 --cfgs wr128x192_s16_d4_l2,wr128x192_abl7_nostore,wr128x192_abl8_plainst,wr128x192_abl9_empty,wr128x192_abl3_mfma --rounds 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02q_ab_store.txt
