#!/usr/bin/env python3
"""Cold-weights forward with and without the cross-layer prefetch experiment (mixq_gemm_hint_next_weights, tuning library only; forms by MIXQ_PF_MODE:
1 scalar-cache touches, 2 plain vector touches, 6 vector touches with the nt hint, 0 none), alternated:
`--copies` layer copies (> 256 MB of weight images) in rotation inside ONE hipGraph per arm, the arms replayed A B A B ..., median per arm.
(bench.py's first form of this comparison timed the arms one after the other and showed a 2 us order effect with the prefetch compiled OUT.)
  python tools/cold_prefetch_ab.py [--shape 4096,11008] [--rows 512] [--copies 8] [--rounds 9]"""
import argparse
import os
import sys

import numpy as np
import torch
os.environ.setdefault("MIXQ_TUNING_LIB", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="4096,11008")
ap.add_argument("--rows", type=int, default=512)
ap.add_argument("--copies", type=int, default=8)
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--bit", type=int, default=8)
args = ap.parse_args()
K, N = (int(v) for v in args.shape.split(","))
dev = "cuda"
bench.K, bench.N = K, N
cols, base, pristine = bench.make_batches(8, args.rows, dev, 0, Kc=K)
layers, cache = [], None
for c in range(args.copies):
    _, cache, lc = bench.build_layer(dev, args.rows, seed=c, bit=args.bit, cache=cache, shape=(K, N))
    for _ in range(3):
        lc(base.clone(), None, True)
    layers.append(lc)
torch.cuda.synchronize()
from mixq_amd import _capi  # noqa: E402
lib = _capi.load()


def forward(i, x, hint):
    if hint:                                                   # consumed by the next weights-in-registers launch of this thread: layer i's GEMM
        nxt = layers[(i + 1) % len(layers)]._wpk
        assert lib.mixq_gemm_hint_next_weights(nxt.data_ptr(), nxt.numel()) == 0
    return layers[i](x, None, True)


steps = args.copies * 4
side = torch.cuda.Stream()
graphs = {}
with torch.cuda.stream(side):
    for arm in ("no hint", "hint"):
        pristine.copy_(base.unsqueeze(0).expand_as(pristine))
        for i in range(args.copies):
            forward(i, pristine[i % pristine.shape[0]], arm == "hint")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(steps):
                forward(i % args.copies, pristine[i % pristine.shape[0]], arm == "hint")
        torch.cuda.synchronize()
        graphs[arm] = g
    times = {a: [] for a in graphs}
    for r in range(args.rounds + 2):
        for arm, g in graphs.items():
            pristine.copy_(base.unsqueeze(0).expand_as(pristine))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side); g.replay(); e1.record(side)
            torch.cuda.synchronize()
            if r >= 2:
                times[arm].append(e0.elapsed_time(e1) * 1e3 / steps)
wbytes = int(layers[0]._wpk.numel())
print(f"{args.rows} x {K} -> {N} W{args.bit}A{args.bit}, {args.copies} layer copies in rotation ({args.copies * wbytes >> 20} MB of weight images), MIXQ_PF_MODE={os.environ.get('MIXQ_PF_MODE', '(default)')}: "
      + " | ".join(f"{a}: {np.median(t):.2f} us per forward (min {min(t):.2f})" for a, t in times.items()))
