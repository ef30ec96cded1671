#!/bin/bash
mkdir -p gpurun_out
export MIXQ_TUNING_LIB=1
{
python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
from mixq_amd import _capi, mixlib
lib = _capi.load(); names = _capi.gemm_config_names()
g = torch.Generator().manual_seed(0)
def nib(R, K, lo):
    v = torch.randint(lo, 8, (R, K), generator=g, dtype=torch.int8); u = (v & 0xF).to(torch.uint8)
    return (u[:, 0::2] | (u[:, 1::2] << 4)).contiguous().cuda()
bad = 0
for (M, N, K, n_out) in [(512, 11008, 4096, 128), (200, 328, 512, 143), (33, 256, 128, 0), (512, 4096, 11008, 110), (130, 200, 1024, 64), (96, 320, 256, 3), (512, 12288, 4096, 41)]:
    qx, qw = nib(M, K, -7), nib(N, K, -8)
    sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().cuda(); sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().cuda()
    xo = wo = None
    if n_out:
        pad = (n_out + 15) // 16 * 16
        xo = torch.randn((M, pad), generator=g).half().cuda()[:, :n_out]; wo = torch.randn((N, pad), generator=g).half().cuda()[:, :n_out]
    bias = torch.randn(N, generator=g).half().cuda()
    x6, w6 = mixlib.PackOperand(qx, 4), mixlib.PackOperand(qw, 3)
    lib.mixq_gemm_set_config(-1)
    want = mixlib.FusedLinear(x6, w6, sx, sw, xo, wo, n_out, bias, M, N, K, bit=4)
    for nm in ("wr128x192_f6r_d2", "wr128x192_f6r_d3", "wr64x128_f6r_d3", "wr64x256_f6r_d3"):
        assert lib.mixq_gemm_set_config(names.index(nm)) == 0
        for rep in range(3):
            got = mixlib.FusedLinear(x6, w6, sx, sw, xo, wo, n_out, bias, M, N, K, bit=4)
        torch.cuda.synchronize()
        ok = torch.equal(want, got); bad += not ok
        print(f"{M}x{N}x{K} n_out={n_out} {nm}: {'bit-identical' if ok else 'MISMATCH %d' % int((want != got).sum())}", flush=True)
    lib.mixq_gemm_set_config(-1)
print("FAILS", bad)
PY
for shp in 512x11008x4096 512x4096x11008 512x4096x4096; do
timeout 300 python tools/ab_gemm.py --shape $shp --bit 4 --f6 --nout 128 --rounds 20 --cfgs wr128x192_s16_d4_l2,wr128x192_f6r_d2,wr128x192_f6r_d3,wr64x128_s16_d4_l2,wr64x128_f6r_d3,wr64x256_s16_d4_l2,wr64x256_f6r_d3
done
} > gpurun_out/r03_f6r.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03_f6r.txt | tail -60
