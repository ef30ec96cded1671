#!/usr/bin/env python3
"""SURVEY 8d's remaining secondary timings at the metric shape (512 x 4096 -> 11008, W8A8O16, 1 % outlier columns):
  bias on          steady state with a bias vector (fused into the GEMM epilogue), one hipGraph of 100 forwards;
  warm-up calls    wall time of forward 1 (discovers the outlier columns: detection, weight-column dequantisation, re-quantise; host
                   syncs) and forward 2 (freezes), eager;
  misprediction    a NEW outlier column appears at call 2 (the reference's cache.stop = 2 still lets it in): wall time of that call.
Prints microseconds; -> profiles/r04_secondary_timings.txt"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import MixLibCache, MixLinear_GEMM
import bench  # the measurement protocol (clock conditioning in front of the timed replay)

M, K, N = 512, 4096, 11008
dev = "cuda"
torch.manual_seed(0)
cols = torch.randperm(K, generator=torch.Generator().manual_seed(1))[:41]
base = torch.randn(M, K, generator=torch.Generator().manual_seed(100)).half()
base[:, cols] *= 20


def graph_time(layer, steps=100):
    xs = base.to(dev).unsqueeze(0).repeat(steps, 1, 1).contiguous()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for i in range(5):
            layer(xs[i], None, True)
        xs.copy_(base.to(dev).unsqueeze(0).expand_as(xs))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(steps):
                layer(xs[i], None, True)
        torch.cuda.synchronize()
        bd = base.to(dev)
        ms, _, _ = bench.conditioned_replay(g, side, restore=lambda: xs.copy_(bd.unsqueeze(0).expand_as(xs)))
    return ms * 1e3 / steps


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for bias in (False, True):
    lin = torch.nn.Linear(K, N, bias=bias).half()
    cache = MixLibCache(M, device=dev)
    layer = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=dev)
    layer(torch.randn(M, K).half().to(dev), None, True)                     # (library / allocator warm-up on another layer state is not what we time)
    layer = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=dev)
    t1 = wall(lambda: layer(base.clone().to(dev), None, True))
    t2 = wall(lambda: layer(base.clone().to(dev), None, True))
    assert layer.add_outliers is False and set(cols.tolist()) <= set(layer.ind.cpu().tolist())
    print(f"bias={bias}: forward 1 (discovers {layer.ind.numel()} outlier columns) {t1:8.1f} us wall, forward 2 (freezes) {t2:8.1f} us wall, "
          f"steady state {graph_time(layer):6.2f} us per forward")
# misprediction: a column the first call did not see turns up at call 2
lin = torch.nn.Linear(K, N, bias=False).half()
cache = MixLibCache(M, device=dev)
layer = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=dev)
layer(base.clone().to(dev), None, True)
n1 = layer.ind.numel()
x2 = base.clone()
newc = [c for c in range(K) if c not in set(cols.tolist())][:3]
x2[:, newc] *= 25
t = wall(lambda: layer(x2.to(dev), None, True))
print(f"misprediction at call 2: {layer.ind.numel() - n1} new columns appended, {t:8.1f} us wall (detect + append + re-quantise + GEMM, host syncs); "
      f"steady state afterwards {graph_time(layer):6.2f} us per forward with {layer.ind.numel()} columns")
