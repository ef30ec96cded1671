#!/usr/bin/env python3
"""Interleaved A/B of the fused GEMM between two BUILDS of the library (e.g. before / after a kernel change): each .so is loaded
through its own ctypes handle (separate globals), gets one hipGraph of `--launches` launches of mixq_gemm_i8_fused on the same
operands, and the graphs are replayed round-robin (cdna_hip_programming.md 5.4 rules 13 / 24).  Also checks that the two builds
produce bit-identical outputs.  Development tool: the second library never ships."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True, help="comma list of name=path (path relative to the repo root)")
ap.add_argument("--shapes", default="512x11008x4096")
ap.add_argument("--nouts", default="41")
ap.add_argument("--rounds", type=int, default=30)
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--bit", type=int, default=8, help="4: mixq_gemm_i4_fused on FP6-coded operands (the W4A4 form of the weights-in-registers kernels)")
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_capi.load()
libs = []
for item in args.libs.split(","):
    nm, path = item.split("=")
    h = C.CDLL(os.path.join(root, path))
    h.mixq_gemm_i8_fused.argtypes = _capi.SIGNATURES["mixq_gemm_i8_fused"]
    h.mixq_gemm_i8_fused.restype = C.c_int
    h.mixq_gemm_i4_fused.argtypes = _capi.SIGNATURES["mixq_gemm_i4_fused"]
    h.mixq_gemm_i4_fused.restype = C.c_int
    libs.append((nm, h))
dev = "cuda"
side = torch.cuda.Stream()
for shp in args.shapes.split(","):
    M, N, K = (int(v) for v in shp.split("x"))
    for nout in (int(v) for v in args.nouts.split(",")):
        g = torch.Generator().manual_seed(0)
        xf = torch.randn(M, K, generator=g)
        qx = torch.round(xf / (xf.abs().amax(dim=1, keepdim=True) / 127)).to(torch.int8).to(dev)
        qw = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
        sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
        sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
        xp, wp = mixlib.PackOperand(qx, 1), mixlib.PackOperand(qw, 2)
        if args.bit == 4:
            from mixq_amd.linear import pack_to_i4
            wp = mixlib.PackOperand(pack_to_i4(torch.randint(-8, 8, (N, K), generator=g, dtype=torch.int8)).to(dev), 3)
            xp = mixlib.PackOperand(pack_to_i4(torch.randint(-7, 8, (M, K), generator=g, dtype=torch.int8)).to(dev), 4)
        lay = (_capi.X_PACKED | _capi.W_F16X64) if args.bit == 8 else _capi.XW_F6X128
        pad = (max(nout, 1) + 15) // 16 * 16
        xo = torch.randn((M, pad), device=dev).half()
        wo = torch.randn((N, pad), device=dev).half()
        outs, graphs = {}, []
        with torch.cuda.stream(side):
            st = side.cuda_stream
            for nm, h in libs:
                out = torch.zeros((M, N), dtype=torch.float16, device=dev)
                outs[nm] = out
                fn = h.mixq_gemm_i8_fused if args.bit == 8 else h.mixq_gemm_i4_fused
                run = lambda fn=fn, out=out: fn(xp.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(),
                                                xo.data_ptr() if nout else None, pad, wo.data_ptr() if nout else None, pad, nout, None, None, 0,
                                                None, out.data_ptr(), N, M, N, K, 0, lay, st)
                for _ in range(3):
                    assert run() == 0
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    for _ in range(args.launches):
                        run()
                torch.cuda.synchronize()
                graphs.append((nm, gr))
            times = {nm: [] for nm, _ in graphs}
            for r in range(args.rounds + 2):
                for nm, gr in graphs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side); gr.replay(); e1.record(side)
                    torch.cuda.synchronize()
                    if r >= 2:
                        times[nm].append(e0.elapsed_time(e1) * 1e3 / args.launches)
        first = outs[libs[0][0]]
        same = all(torch.equal(first, o) for o in outs.values())
        flops = 2.0 * M * N * K
        print(f"{shp} n_out={nout}: {args.rounds} interleaved rounds x {args.launches} launches; us per launch median / min; outputs bit-identical: {same}")
        for nm, _ in graphs:
            t = np.array(times[nm])
            print(f"  {nm:12s} {np.median(t):8.2f} {t.min():8.2f}   {flops / np.median(t) / 1e6:7.1f} TOPS ({100 * flops / np.median(t) / 1e6 / 5033:4.1f} %)", flush=True)
