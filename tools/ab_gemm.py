#!/usr/bin/env python3
"""Interleaved A/B timing of GEMM tile configurations (cdna_hip_programming.md 5.4 rules 13/24: deltas under 10 % need within-probe
interleaved rounds).  Every configuration gets ONE hipGraph of `--launches` launches on the same operands; the graphs are replayed
round-robin for `--rounds` rounds, each replay timed with HIP events, and the per-configuration median / min over the rounds is
reported - drift of clocks and temperature hits all configurations alike."""
import argparse, os, sys
import numpy as np
import os
import torch
os.environ.setdefault("MIXQ_TUNING_LIB", "1")      # ablation kernels / trace stamps / probe knobs live in the tools build (make tuning)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="512x11008x4096")
ap.add_argument("--cfgs", required=True, help="comma list of config NAMES")
ap.add_argument("--rounds", type=int, default=30)
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--nout", type=int, default=41)
ap.add_argument("--bit", type=int, default=8)
ap.add_argument("--f6", action="store_true", help="bit 4 with both operands as FP6 codes (MIXQ_FMT_F6X128): the FP6-pipe form of the wr kernels")
ap.add_argument("--cold", type=int, default=0, help="rotate the launches of a graph over this many copies of the weight image (8 x 45 MB > the 256 MB "
                "memory-side cache: every launch streams its weights from HBM, as the layers of a model do); 0: one copy, resident after the first launch")
ap.add_argument("--gms", default="0", help="comma list of M-tile group sizes of the weights-in-registers kernels' tile order (0 = automatic)")
args = ap.parse_args()
dev = "cuda"
lib = _capi.load()
names = _capi.gemm_config_names()
M, N, K = (int(v) for v in args.shape.split("x"))
g = torch.Generator().manual_seed(0)
KB = K if args.bit == 8 else K // 2
if os.environ.get("AB_UNIFORM_X"):
    qx = torch.randint(-127, 128, (M, KB), generator=g, dtype=torch.int8).to(dev)
else:                                                   # the bench's operand statistics: activations quantised from N(0,1) rows (|q| <= 127,
    xf = torch.randn(M, KB, generator=g)                # typically ~30): uniform +-127 bytes toggle more and cost clock (power limit)
    qx = torch.round(xf / (xf.abs().amax(dim=1, keepdim=True) / 127)).to(torch.int8).to(dev)
qw = torch.randint(-127, 128, (N, KB), generator=g, dtype=torch.int8).to(dev)
if args.bit == 4:
    qx, qw = qx.view(torch.uint8), qw.view(torch.uint8)
sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
xp = mixlib.PackOperand(qx, 4 if args.f6 else 1)
wp = {1: mixlib.PackOperand(qw, 1), 2: mixlib.PackOperand(qw, 3 if args.f6 else 2)}
ncopy = max(1, args.cold)
wcopies = {k: [v] + [mixlib.set_fmt(v.clone(), mixlib.fmt_of(v)) for _ in range(ncopy - 1)] for k, v in wp.items()}
xo = wo = None
if args.nout:
    pad = (args.nout + 15) // 16 * 16
    xo = torch.randn((M, pad), device=dev).half()[:, :args.nout]
    wo = torch.randn((N, pad), device=dev).half()[:, :args.nout]
out = torch.empty((M, N), dtype=torch.float16, device=dev)
if any(nm.startswith("sk") or nm.endswith("_k2") for nm in args.cfgs.split(",")):
    _capi.ensure_workspace(dev)                        # stream-K forms hand partial tiles over through a registered workspace
side = torch.cuda.Stream()
graphs = []
with torch.cuda.stream(side):
    for nm, gm in [(nm, int(g)) for nm in args.cfgs.split(",") for g in args.gms.split(",")]:
        c = names.index(nm)
        w = wp[2 if nm.startswith("wr") else 1]
        assert lib.mixq_gemm_set_config(c) == 0
        assert lib.mixq_gemm_set_krot(gm << 16) == 0
        if gm:
            nm = f"{nm}_gm{gm}"
        xa = xp if (nm.startswith("wr") or not args.f6) else mixlib.PackOperand(qx, 1)     # (the LDS-staged kernels take nibbles in P16X64)
        ws = wcopies[2 if nm.startswith("wr") else 1]
        run = lambda i=0, xa=xa, ws=ws: mixlib.FusedLinear(xa, ws[i % len(ws)], sx, sw, xo, wo, args.nout, None, M, N, K, bit=args.bit, out=out)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for i in range(args.launches):
                run(i)
        torch.cuda.synchronize()
        graphs.append((nm, gr))
    lib.mixq_gemm_set_config(-1)
    lib.mixq_gemm_set_krot(0)
    times = {nm: [] for nm, _ in graphs}
    for r in range(args.rounds + 2):
        for nm, gr in graphs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side); gr.replay(); e1.record(side)
            torch.cuda.synchronize()
            if r >= 2:
                times[nm].append(e0.elapsed_time(e1) * 1e3 / args.launches)
flops = 2.0 * M * N * K
print(f"{args.shape} bit={args.bit} n_out={args.nout}{' COLD weights (' + str(ncopy) + ' copies in rotation)' if ncopy > 1 else ''}: {args.rounds} interleaved rounds x {args.launches} launches; us per launch median / min / max")
for nm, _ in graphs:
    t = np.array(times[nm])
    print(f"  {nm:28s} {np.median(t):7.2f} {t.min():7.2f} {t.max():7.2f}   {flops / np.median(t) / 1e6:7.1f} TOPS ({100 * flops / np.median(t) / 1e6 / 5033:4.1f} %)", flush=True)
