#!/usr/bin/env python3
"""Prefill-sized GEMM + fused epilogue (no outlier columns): every candidate tiling of this library against its automatic pick and the
vendor int8 GEMM (torch._int_mm = hipBLASLt), interleaved rounds, us per launch (median).  Shapes: M tokens x the reference's published
layers (bench/README.md:6-12: 7B / 13B / 70B at M = 2048; examples/benchW8A8.ipynb).   python tools/prefill_sweep.py [--tokens 1024,2048,4096,8192]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", default="1024,2048,4096,8192")
ap.add_argument("--layers", default="11008x4096,4096x11008,12288x4096,14336x4096,13824x5120,5120x13824,12288x8192,28672x8192,8192x28672")
ap.add_argument("--cfgs", default="256x256_w4x2_s5_l0,wr128x192_s16_d4_l2,wr128x256_s16_d3_l2", help="tilings to force, by name or id")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--bit", type=int, default=8, help="4: W4A4 with both operands as FP6 codes (the FP6-pipe form of the weights-in-registers kernels)")
args = ap.parse_args()
dev = "cuda"
lib = _capi.load()
if os.environ.get("MIXQ_TUNING_LIB") == "1":
    _capi.ensure_workspace(dev)                # (the pairwise split-K form, tuning library, hands partial tiles through it)
names = _capi.gemm_config_names()
print(_capi.device_info())
cfgs = [int(c) if c.lstrip("-").isdigit() else names.index(c) for c in args.cfgs.split(",")]


def graph_of(fn, n):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
        torch.cuda.synchronize()
    return g, st


def time_graph(g, st, n):
    with torch.cuda.stream(st):
        g.replay()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        g.replay()
        e.record()
        torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (2 * n)


print("us per launch (median of interleaved rounds); * = the tiling the model picks; auto = that tiling + the N split where it is priced to pay; outputs of all arms bit-identical")
for layer in args.layers.split(","):
    N, K = (int(v) for v in layer.split("x"))
    g0 = torch.Generator(device="cpu").manual_seed(0)
    qw = torch.randint(-127, 128, (N, K), generator=g0, dtype=torch.int8).to(dev)
    if args.bit == 4:
        from mixq_amd.linear import pack_to_i4
        wf = mixlib.PackOperand(pack_to_i4(torch.randint(-8, 8, (N, K), generator=g0, dtype=torch.int8)).to(dev), 3)
    else:
        wf = mixlib.PackOperand(qw, 2)
    sw = (torch.rand(1, N, generator=g0) * 0.01 + 0.001).half().to(dev)
    for M in (int(v) for v in args.tokens.split(",")):
        qx = torch.randint(-127, 128, (M, K), generator=g0, dtype=torch.int8).to(dev)
        xp = mixlib.PackOperand(pack_to_i4(torch.randint(-7, 8, (M, K), generator=g0, dtype=torch.int8)).to(dev), 4) if args.bit == 4 else mixlib.PackOperand(qx, 1)
        sx = (torch.rand(M, 1, generator=g0) * 0.01 + 0.001).half().to(dev)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        n = max(2, min(20, int(4000 / max(1.0, 2e-9 * M * N * K / 1.8))))
        auto = lib.mixq_gemm_pick_config_fmt(M, N, K, args.bit, 3 if args.bit == 4 else 2)
        arms = {}
        ref = None
        # -1: the automatic choice (tiling by the model, N split when the last round of tiles is partial); -2: the same without the split
        for c in cfgs + [-2, -1]:
            assert lib.mixq_gemm_set_config(c) == 0
            nm = names[c] if c >= 0 else ("auto" if c == -1 else "auto-1launch")
            try:
                f = lambda: mixlib.FusedLinear(xp, wf, sx, sw, None, None, 0, None, M, N, K, bit=args.bit, out=out)
                out.zero_()
                f()
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(ref, out):
                    print(f"  !! {nm} differs from {names[cfgs[0]]} at {M}x{N}x{K}: max {float((ref.float() - out.float()).abs().max())}")
                arms[nm] = graph_of(f, n)
            except Exception as ex:
                print(f"  {M}x{N}x{K} {nm}: {ex}")
        lib.mixq_gemm_set_config(-1)
        qwt = qw.t()
        arms["int_mm"] = graph_of(lambda: torch._int_mm(qx, qwt), n)
        res = {k: [] for k in arms}
        for r in range(args.rounds):
            for k, (g, st) in arms.items():
                res[k].append(time_graph(g, st, n))
        med = {k: statistics.median(v) for k, v in res.items()}
        ours = {k: v for k, v in med.items() if k != "int_mm"}
        best = min(ours, key=ours.get)
        line = "  ".join(f"{k}{'*' if k == names[auto] else ''} {v:8.2f}" for k, v in med.items())
        flag = "" if ours["auto"] <= 1.01 * ours[best] else f"   <-- auto loses {100 * (ours['auto'] / ours[best] - 1):.1f} %"
        print(f"{M:5d} x {K:5d} -> {N:5d}: {line}  | auto {2e-6 * M * N * K / ours['auto']:7.1f} TOPS, vendor / auto = {med['int_mm'] / ours['auto']:.3f}{flag}", flush=True)
        del arms
