#!/usr/bin/env python3
"""Timing probe for the one-launch form (NOTEBOOK.md round 6): quantise + GEMM as TWO launches (the shipped forward: mixq_quant_fused_masked, then the fused GEMM on the
metric tiling) against ONE launch of the tuning configuration wr128x192_p77_one_launch, whose workgroups run a stand-in quantise phase over the same fp16 rows (write-through
stores into a scratch image, one agent-scope add per row on its M tile's counter, the loader waves polling that counter) in front of the unchanged GEMM.  Both arms as hipGraphs
of `--steps` forwards, replayed alternately; the GEMM reads the same pre-quantised operands in both arms and must produce the same bytes.
  MIXQ_TUNING_LIB=1 python tools/fuse_probe.py [--shape 512x11008x4096] [--nout 41]"""
import argparse
import os
import sys

import numpy as np
import torch
os.environ.setdefault("MIXQ_TUNING_LIB", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402
from mixq_amd.linear import kept_outlier_map  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="512x11008x4096")
ap.add_argument("--nout", type=int, default=41)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=15)
args = ap.parse_args()
M, N, K = (int(v) for v in args.shape.split("x"))
dev = "cuda"
lib = _capi.load()
names = _capi.gemm_config_names()
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).half()
cols = torch.sort(torch.randperm(K, generator=g)[:args.nout]).values.to(torch.int32)
x[:, cols.long()] *= 20
x = x.to(dev)
ind = cols.to(dev)
kept = kept_outlier_map(ind, K)                          # the frozen layer's route: mixq_quant_fused_masked, one memory round trip
qw = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
wp = mixlib.PackOperand(qw, 2)
sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
pad = (max(args.nout, 1) + 15) // 16 * 16
wo = torch.randn((N, pad), generator=g).half().to(dev)[:, :args.nout]
x_scale = torch.zeros((M, 1), dtype=torch.float16, device=dev)
xs = [x.clone() for _ in range(args.steps)]
pristine = [t.clone() for t in xs]
scratch = torch.zeros((M, K), dtype=torch.uint8, device=dev)
counters = torch.zeros(64, dtype=torch.int32, device=dev)
out = torch.empty((M, N), dtype=torch.float16, device=dev)


def two_launches(i):
    q, xo = mixlib.QuantFused(xs[i], ind, x_scale, 8, 6.0, fmt=1, col_mask=kept)
    return mixlib.FusedLinear(q, wp, x_scale, sw, xo, wo, args.nout, None, M, N, K, bit=8, out=out)


q0, xo0 = mixlib.QuantFused(xs[0].clone(), ind, x_scale, 8, 6.0, fmt=1, col_mask=kept)      # the operands the one-launch arm's GEMM reads (its quantise phase is a stand-in)
sx0 = x_scale.clone()


def one_launch(i):
    return mixlib.FusedLinear(q0, wp, sx0, sw, xo0, wo, args.nout, None, M, N, K, bit=8, out=out)


side = torch.cuda.Stream()
graphs = {}
with torch.cuda.stream(side):
    for arm, fn, cfg, probe in (("two launches (quantise, GEMM)", two_launches, "wr128x192_s16_d4_l2", False), ("GEMM alone", one_launch, "wr128x192_s16_d4_l2", False),
                                ("one launch (stand-in quantise phase + GEMM)", one_launch, "wr128x192_p77_one_launch", True)):
        assert lib.mixq_gemm_set_config(names.index(cfg)) == 0
        assert lib.mixq_gemm_set_fuse_probe(pristine[0].data_ptr() if probe else None, scratch.data_ptr() if probe else None, counters.data_ptr() if probe else None, K if probe else 0) == 0
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        ref = out.clone()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for i in range(args.steps):
                fn(i)
        torch.cuda.synchronize()
        graphs[arm] = (gr, ref)
    lib.mixq_gemm_set_fuse_probe(None, None, None, 0)
    lib.mixq_gemm_set_config(-1)
    times = {a: [] for a in graphs}
    for r in range(args.rounds + 2):
        for arm, (gr, _) in graphs.items():
            for t, p in zip(xs, pristine):
                t.copy_(p)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side); gr.replay(); e1.record(side)
            torch.cuda.synchronize()
            if r >= 2:
                times[arm].append(e0.elapsed_time(e1) * 1e3 / args.steps)
same = torch.equal(graphs["GEMM alone"][1], graphs["one launch (stand-in quantise phase + GEMM)"][1])
print(f"{args.shape} W8A8, {args.nout} outlier columns, {args.steps} forwards per graph, {args.rounds} alternated replays; us per forward median / min; counters left at {int(counters.abs().sum())} (0 = every tile reset);"
      f" GEMM output of the one-launch arm identical to the GEMM's: {same}")
for arm, t in times.items():
    print(f"  {arm:48s} {np.median(t):8.2f} {min(t):8.2f}")
