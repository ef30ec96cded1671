#!/usr/bin/env python3
"""Interleaved A/B of the FP6-pipe W4A4 GEMM between two BUILDS whose activation formats differ (old: F6X128, new: R6X128): each .so packs
its own operands, gets one hipGraph of launches, graphs replayed round-robin.  Development tool."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_capi.load()
libs = []
for item in sys.argv[1].split(","):
    nm, path, xfmt = item.split("=")
    h = C.CDLL(os.path.join(root, path))
    h.mixq_gemm_i4_fused.argtypes = _capi.SIGNATURES["mixq_gemm_i4_fused"]; h.mixq_gemm_i4_fused.restype = C.c_int
    h.mixq_pack_operand.argtypes = _capi.SIGNATURES["mixq_pack_operand"]; h.mixq_pack_operand.restype = C.c_int
    libs.append((nm, h, int(xfmt)))
dev = "cuda"
side = torch.cuda.Stream()
for shp in sys.argv[2].split(","):
    M, N, K, nout = (int(v) for v in shp.split("x"))
    g = torch.Generator().manual_seed(0)
    xf = torch.randn(M, K, generator=g)
    vx = torch.round(xf / (xf.abs().amax(dim=1, keepdim=True) / 7)).to(torch.int8)
    vw = torch.randint(-8, 8, (N, K), generator=g, dtype=torch.int8)
    nib = lambda v: (((v & 0xF).to(torch.uint8))[:, 0::2] | (((v & 0xF).to(torch.uint8))[:, 1::2] << 4)).contiguous().to(dev)
    qx, qw = nib(vx), nib(vw)
    sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
    sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
    pad = (max(nout, 1) + 15) // 16 * 16
    xo = torch.randn((M, pad), device=dev).half(); wo = torch.randn((N, pad), device=dev).half()
    graphs, outs = [], {}
    with torch.cuda.stream(side):
        st = side.cuda_stream
        for nm, h, xfmt in libs:
            x6 = torch.empty(((M + 15) // 16 * 16, K * 3 // 4), dtype=torch.uint8, device=dev)
            w6 = torch.empty(((N + 15) // 16 * 16, K * 3 // 4), dtype=torch.uint8, device=dev)
            assert h.mixq_pack_operand(qx.data_ptr(), x6.data_ptr(), M, K // 2, xfmt, st) == 0
            assert h.mixq_pack_operand(qw.data_ptr(), w6.data_ptr(), N, K // 2, 3, st) == 0
            out = torch.zeros((M, N), dtype=torch.float16, device=dev); outs[nm] = out
            run = lambda h=h, x6=x6, w6=w6, out=out: h.mixq_gemm_i4_fused(x6.data_ptr(), w6.data_ptr(), sx.data_ptr(), sw.data_ptr(), xo.data_ptr() if nout else None, pad,
                                                                    wo.data_ptr() if nout else None, pad, nout, None, None, 0, None, out.data_ptr(), N, M, N, K, 0, 16, st)
            for _ in range(3):
                assert run() == 0
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                for _ in range(20):
                    run()
            torch.cuda.synchronize()
            graphs.append((nm, gr, (x6, w6)))
        t = {nm: [] for nm, _, _ in graphs}
        for r in range(32):
            for nm, gr, _ in graphs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side); gr.replay(); e1.record(side); torch.cuda.synchronize()
                if r >= 2: t[nm].append(e0.elapsed_time(e1) * 1e3 / 20)
    names = [nm for nm, _, _ in graphs]
    ref = (vx.to(dev).double() @ vw.to(dev).double().T) * sx.double() * sw.double()
    if nout: ref = ref + xo[:, :nout].double() @ wo[:, :nout].double().T
    for nm in names:
        d = (outs[nm].double() - ref).abs()
        print(f"   {nm}: max |y - fp64 reference| = {float(d.max()):.4g} (|ref| max {float(ref.abs().max()):.4g}); differing from {names[0]}: {int((outs[nm] != outs[names[0]]).sum())}")
    same = all(torch.equal(outs[names[0]], outs[nm]) for nm in names[1:])
    print(f"{shp}: outputs {'bit-identical' if same else 'DIFFER'}; us per launch median / min")
    for nm in names:
        a = np.array(t[nm]); print(f"   {nm:28s} {np.median(a):7.2f} {a.min():7.2f}")
