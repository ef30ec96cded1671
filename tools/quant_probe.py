#!/usr/bin/env python3
"""What the quantise pass's parts cost INSIDE the forward (quantise + GEMM graph, bench.py's protocol; a graph of quantise kernels alone drops
the clocks): the tuning library's probe bits of quant_rows2_kernel (results are garbage by design) - 1 no in-place zeroing, 8 no x_out
stores, 4 no outlier handling at all, 32 no quantise arithmetic.   MIXQ_TUNING_LIB=1 python tools/quant_probe.py [--shape K,N]"""
import argparse
import os
import statistics
import sys

import torch

os.environ.setdefault("MIXQ_TUNING_LIB", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mixq_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=512)
ap.add_argument("--shape", default="4096,11008")
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
bench.K, bench.N = (int(v) for v in args.shape.split(","))
M, dev, steps = args.tokens, torch.device("cuda", 0), 20
lib = _capi.load()
lin, cache, layer = bench.build_layer(dev, M)
cols, base, pristine = bench.make_batches(steps, M, dev, 0)
for _ in range(3):
    layer(base.clone(), None, True)
torch.cuda.synchronize()
assert not layer.add_outliers
st = torch.cuda.Stream()
PROBES = [(0, "full"), (1, "no in-place zeroing of x"), (8, "no x_out stores"), (4, "no outlier handling at all"), (32, "no quantise arithmetic"), (64, "positions loaded, nothing extracted"), (64 | 8, "... and no x_out stores"),
          (4 | 32, "neither outliers nor arithmetic")]
graphs = {}
for dbg, name in PROBES:
    assert lib.mixq_quant_set_config(100 + dbg) == 0
    with torch.cuda.stream(st):
        layer(base.clone(), None, True)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for i in range(steps):
                layer(pristine[i], None, True)
    graphs[name] = gr
lib.mixq_quant_set_config(100)
res = {k: [] for k in graphs}
with torch.cuda.stream(st):
    for r in range(args.rounds):
        for name, gr in graphs.items():
            ms, _, _ = bench.conditioned_replay(gr, st, restore=lambda: pristine.copy_(base.unsqueeze(0).expand_as(pristine)))
            res[name].append(ms * 1e3 / steps)
print(f"forward of the frozen layer with probe bits in the quantise pass (tuning library), {M} x {bench.K} -> {bench.N}, {int(layer.ind.numel())} outlier columns")
full = statistics.median(res["full"])
for name, v in res.items():
    print(f"  {name:34s} median {statistics.median(v):6.2f} us ({statistics.median(v) - full:+.2f})   all: " + " ".join(f"{t:.2f}" for t in v))
