cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "tiling or k_step or wreg or randomized_shapes or outlier_columns_every or fused_linear_vs or full_size" 2>&1 | tail -3
timeout 300 python tools/ab_gemm.py --shape 512x11008x4096 --cfgs 128x192_w2x2_s5_l4,wr128x192_s16_d4_l2,wr128x192_abl12_noramp,wr128x192_s8_d4_l1 --rounds 40 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02x_ab_ramp.txt
timeout 300 python tools/ab_gemm.py --shape 512x8192x8192 --nout 82 --cfgs 128x128_w2x2_s5_l2,wr128x128_s16_d4_l2,wr128x128_s8_d4_l1 --rounds 20 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02x_ab_ramp.txt
timeout 300 python tools/trace_gemm.py --cfgs 29 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02x_ab_ramp.txt
