#!/usr/bin/env python3
"""The forward of the frozen metric layer (quantise pass + GEMM from ONE C call) with the quantise pass building its column mask in the
kernel (mixq_quant_fused) against the kept-mask form (mixq_quant_fused_masked): same-run interleaved A/B of two graphs of 20 forwards
under bench.py's protocol.  (The quantise pass alone cannot be timed this way: a graph of nothing but 5 us kernels lets the part drop
its clocks - 25 us per launch after the first measurement.)
  python tools/time_quant_mask.py [--bit 8] [--shape 4096,11008] [--tokens 512] [--rounds 9]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=512)
ap.add_argument("--shape", default="4096,11008")
ap.add_argument("--bit", type=int, default=8)
ap.add_argument("--rounds", type=int, default=9)
args = ap.parse_args()
bench.K, bench.N = (int(v) for v in args.shape.split(","))
M, dev, steps = args.tokens, torch.device("cuda", 0), 20
lin, cache, layer = bench.build_layer(dev, M, bit=args.bit)
cols, base, pristine = bench.make_batches(steps, M, dev, 0)
for _ in range(3):
    layer(base.clone(), None, True)
torch.cuda.synchronize()
assert not layer.add_outliers
y_ref = layer(base.clone(), None, True)
plan = layer._plan
mask = plan.kept_mask
assert mask is not None
st = torch.cuda.Stream()
graphs = {}
for name, cm in (("mask built in the kernel", None), ("kept mask", mask)):
    plan.kept_mask = cm
    with torch.cuda.stream(st):
        y = layer(base.clone(), None, True)
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref), name
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for i in range(steps):
                layer(pristine[i], None, True)
    graphs[name] = gr
plan.kept_mask = mask
res = {k: [] for k in graphs}
with torch.cuda.stream(st):
    for r in range(args.rounds):
        for name, gr in graphs.items():
            ms, _, _ = bench.conditioned_replay(gr, st, restore=lambda: pristine.copy_(base.unsqueeze(0).expand_as(pristine)))
            res[name].append(ms * 1e3 / steps)
print(f"forward of the frozen layer, W{args.bit}A{args.bit}, {M} x {bench.K} -> {bench.N}, {int(layer.ind.numel())} outlier columns; {steps} forwards per graph, "
      f"{args.rounds} interleaved rounds; outputs bit-identical")
for name, v in res.items():
    print(f"  {name:28s} median {statistics.median(v):6.2f} us  min {min(v):6.2f}   all: " + " ".join(f"{t:.2f}" for t in v))
