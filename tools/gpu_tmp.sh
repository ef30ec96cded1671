cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "wreg or every_k or randomized_shapes or more_than_128 or fused_linear or skinny or device_memory or silu or operator or baseline or full_size" 2>&1 | tail -3
python tools/ab_gemm.py --cfgs 128x192_w2x2_s5_l4,wr128x192_s16_d4_l2,wr128x192_s8_d4_l1 2>&1 | grep -v amdgpu.ids
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,128x192_w2x2_s5_l4 --nout 0 2>&1 | grep -v amdgpu.ids
python tools/ab_gemm.py --cfgs wr128x192_s16_d4_l2,128x192_w2x2_s5_l4 --nout 143 2>&1 | grep -v amdgpu.ids
