cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -k "w8a16" -x 2>&1 | tail -3
timeout 600 python tools/sweep_w8a16.py --shapes 512x11008x4096,512x4096x4096,512x4096x11008 --cfgs auto,w8a16_128x192_s8_d3_l2,w8a16_128x192_abl5_noramp,w8a16_64x128_s12_d4_l2 --rounds 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02y_w8a16.txt
