#!/usr/bin/env python3
"""Llama-2-7b MLP block through the mirrored modules (fused RMSNorm + quantise -> up_proj, gate_proj with the SiLU(*up)
epilogue -> down_proj), one hipGraph of forwards, M = 512: us per block with and without the fused gate * up product."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import FasterTransformerRMSNorm, MixLibCache, MixLinear_GEMM, MixLlamaMLP, fused  # noqa: E402
import bench  # noqa: E402  (the measurement protocol)

M, H, F = 512, 4096, 11008
dev = "cuda"
torch.manual_seed(0)
cache = MixLibCache(M, sigma=6, bit=8, device=dev)
mk = lambda k, n: MixLinear_GEMM.from_linear(torch.nn.Linear(k, n, bias=False).half(), 8, cache=cache, dev=dev)
gate, up, down = mk(H, F), mk(H, F), mk(F, H)
norm = FasterTransformerRMSNorm((torch.rand(H) + 0.5).half().to(dev), 1e-5, cache)
norm.next_layer = up
mlp = MixLlamaMLP(gate, down, up, cache)
g = torch.Generator().manual_seed(1)
cols = torch.randperm(H, generator=g)[:41]
base = torch.randn(M, H, generator=g).half()
base[:, cols] *= 20
base = base.to(dev)


def block(x, fused_mul):
    h = norm(x)
    if fused_mul:
        return mlp(h)
    upo = up(h, cache)
    go = gate.forward_without_preconditionFusedSilu(h, cache)
    go *= upo
    return down(go, None, True)


cfg = mlp.config                         # the model's switches (mixq_amd/config.py); shared with the layers through the cache
cfg.fuse_down_amax = False
cfg.joint_gate_up = False
for _ in range(3):                       # outlier prediction warm-up (host syncs allowed here)
    block(base.clone(), True)
torch.cuda.synchronize()
ya, yb = block(base.clone(), True), block(base.clone(), False)
print("max |fused - two-step| =", float((ya.float() - yb.float()).abs().max()))
steps = 50
cfg.norm_kept_map = False
for fused_mul, amax, joint, kept in ((False, False, False, False), (True, False, False, False), (True, True, False, False), (True, True, True, False), (True, True, True, True)):
    cfg.fuse_down_amax, cfg.joint_gate_up, cfg.norm_kept_map = amax, joint, kept
    if joint:
        yj = block(base.clone(), True)
        print("joint gate / up launch: max |joint - two launches| =", float((yj.float() - ya.float()).abs().max()), "(bit-identical)" if torch.equal(yj, ya) else "(DIFFERENT)")
    xs = base.unsqueeze(0).repeat(steps, 1, 1).contiguous()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for i in range(steps):
                block(xs[i], fused_mul)
        torch.cuda.synchronize()
        ms, first_ms, _ = bench.conditioned_replay(gr, side, restore=lambda: xs.copy_(base.unsqueeze(0).expand_as(xs)))
        us = ms * 1e3 / steps
    flops = 2.0 * M * (2 * H * F + F * H)
    print(f"norm + MLP block, gate*up {('in ONE joint gate / up launch' if joint else 'in the gate epilogue') if fused_mul else 'as a separate pass'}{', down_proj row maxima from the same epilogue' if amax else ''}{', norm with the kept outlier map' if kept else ''}: {us:7.1f} us  "
          f"({flops / us / 1e6:6.0f} effective TFLOPS; first replay {first_ms * 1e3 / steps:7.1f} us)")
