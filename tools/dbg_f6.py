#!/usr/bin/env python3
"""W4A4 on the FP6 matrix pipe (MIXQ_FMT_F6X128 operands) against the int8-expansion W4A4 path on the same operands: pack / unpack
round trip, the quantiser's F6 output, bit-identity of the GEMM for every tiling that exists in the FP6 form, and us per launch.
Development tool."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402
from mixq_amd._capi import FMT_F6X128, FMT_R6X128, FMT_P16X64, FMT_PLAIN  # noqa: E402

dev = "cuda"
lib = _capi.load()
names = _capi.gemm_config_names()
g = torch.Generator().manual_seed(0)
fails = 0


def nibbles(R, K, lo=-8):
    v = torch.randint(lo, 8, (R, K), generator=g, dtype=torch.int8)
    u = (v & 0xF).to(torch.uint8)
    return v, (u[:, 0::2] | (u[:, 1::2] << 4)).contiguous()


# 1. pack / unpack
for R, K in [(16, 128), (37, 256), (500, 4096)]:
    v, p = nibbles(R, K)
    img = mixlib.PackOperand(p.to(dev), FMT_F6X128)
    img2 = mixlib.PackOperand(p.to(dev), FMT_R6X128)
    back = mixlib.UnpackOperand(img, R)
    ok = torch.equal(back.cpu(), p) and torch.equal(mixlib.UnpackOperand(img2, R).cpu(), p)
    print(f"pack/unpack {R}x{K}: image {tuple(img.shape)} {'ok' if ok else 'MISMATCH'}")
    fails += not ok
# 2. quantiser
for M, K, n in [(512, 4096, 41), (37, 256, 0), (100, 11008 // 128 * 128, 17)]:
    x = torch.randn(M, K, generator=g).half()
    ind = torch.randperm(K, generator=g)[:n].to(torch.int32).to(dev) if n else None
    xs1, xs2 = torch.zeros(M, dtype=torch.float16, device=dev), torch.zeros(M, dtype=torch.float16, device=dev)
    q_plain, xo1 = mixlib.QuantFused(x.clone().to(dev), ind, xs1, 4, 6.0, fmt=FMT_PLAIN)
    q_f6, xo2 = mixlib.QuantFused(x.clone().to(dev), ind, xs2, 4, 6.0, fmt=FMT_R6X128)
    back = mixlib.UnpackOperand(q_f6, M)
    ok = torch.equal(back, q_plain) and torch.equal(xs1, xs2) and (n == 0 or torch.equal(xo1, xo2))
    print(f"quantiser {M}x{K} n_out={n}: F6 image {tuple(q_f6.shape)} {'ok' if ok else 'MISMATCH'}")
    fails += not ok

# 3. GEMM bit-identity and timing
f6_cfgs = [nm for nm in names if nm.startswith("wr") and nm.split("_")[0] in ("wr128x192", "wr128x128", "wr128x256", "wr64x128", "wr64x192", "wr64x256")
           and "_s16_" in nm and "_l2" in nm and "abl" not in nm and "_p" not in nm and "d5" not in nm and "l4" not in nm]
f6_cfgs = [nm for nm in f6_cfgs if nm in ("wr128x192_s16_d4_l2", "wr128x128_s16_d4_l2", "wr64x128_s16_d4_l2", "wr64x192_s16_d4_l2", "wr64x256_s16_d4_l2")]
shapes = [(512, 11008, 4096, 41), (512, 4096, 4096, 0), (500, 4100, 1024, 19), (512, 4096, 11008, 110), (33, 256, 128, 0), (512, 12288, 4096, 128)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1].split(",")]
for M, N, K, n_out in shapes:
    _, qx = nibbles(M, K, -7)
    _, qw = nibbles(N, K)
    qx, qw = qx.to(dev), qw.to(dev)
    sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
    sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
    xo = wo = None
    if n_out:
        pad = (n_out + 15) // 16 * 16
        xo = torch.randn((M, pad), generator=g).half().to(dev)[:, :n_out]
        wo = torch.randn((N, pad), generator=g).half().to(dev)[:, :n_out]
    bias = torch.randn(N, generator=g).half().to(dev)
    xp, wp = mixlib.PackOperand(qx, FMT_P16X64), mixlib.PackOperand(qw, FMT_P16X64)
    x6, w6 = mixlib.PackOperand(qx, FMT_R6X128), mixlib.PackOperand(qw, FMT_F6X128)
    lib.mixq_gemm_set_config(-1)
    want = mixlib.FusedLinear(xp, wp, sx, sw, xo, wo, n_out, bias, M, N, K, bit=4)
    auto = mixlib.FusedLinear(x6, w6, sx, sw, xo, wo, n_out, bias, M, N, K, bit=4)
    torch.cuda.synchronize()
    ok = torch.equal(want, auto)
    print(f"GEMM {M}x{N}x{K} n_out={n_out}: automatic FP6 tiling {'bit-identical' if ok else 'MISMATCH %d' % int((want != auto).sum())}", flush=True)
    fails += not ok
    for nm in f6_cfgs:
        assert lib.mixq_gemm_set_config(names.index(nm)) == 0
        got = mixlib.FusedLinear(x6, w6, sx, sw, xo, wo, n_out, bias, M, N, K, bit=4)
        torch.cuda.synchronize()
        ok = torch.equal(want, got)
        if not ok:
            d = (want != got)
            print(f"   {nm}: MISMATCH {int(d.sum())} of {M * N}; rows {d.any(1).sum().item()} cols {d.any(0).sum().item()} max|d| {float((want.float() - got.float()).abs().max()):.4g}")
        fails += not ok
    lib.mixq_gemm_set_config(-1)
    if M >= 256:
        side = torch.cuda.Stream()
        res = []
        with torch.cuda.stream(side):
            runs = [("int8-expansion (auto)", -1, xp, wp)] + [("fp6 " + nm, names.index(nm), x6, w6) for nm in f6_cfgs]
            graphs = []
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            for label, c, a, b in runs:
                lib.mixq_gemm_set_config(c)
                f = lambda: mixlib.FusedLinear(a, b, sx, sw, xo, wo, n_out, bias, M, N, K, bit=4, out=out)
                f(); torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    for _ in range(20):
                        f()
                torch.cuda.synchronize()
                graphs.append((label, gr))
            lib.mixq_gemm_set_config(-1)
            t = {l: [] for l, _ in graphs}
            for r in range(12):
                for l, gr in graphs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side); gr.replay(); e1.record(side); torch.cuda.synchronize()
                    if r >= 2:
                        t[l].append(e0.elapsed_time(e1) * 1e3 / 20)
        for l, _ in graphs:
            m = float(np.median(t[l]))
            print(f"   {l:36s} {m:7.2f} us  {2.0 * M * N * K / m / 1e6:7.0f} TOPS")
print("FAILS", fails)
sys.exit(1 if fails else 0)
