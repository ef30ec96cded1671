#!/usr/bin/env python3
"""Time the RMSNorm kernels (plain and fused with extract + quantise) by graph replay over rotating inputs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import mixlib
from tools.sweep_gemm import time_graph

dev = "cuda"
for (M, K) in [(512, 4096), (512, 8192), (16, 4096)]:
    x = torch.randn(32, M, K, device=dev).half()
    w = (torch.rand(K, device=dev) + 0.5).half()
    out = torch.empty(M, K, dtype=torch.float16, device=dev)
    xs = torch.zeros(M, 1, dtype=torch.float16, device=dev)
    i = [0]
    def plain():
        mixlib.layernorm_forward_cuda(x[i[0] % 32], w, out, 1e-5); i[0] += 1
    us = time_graph(plain, 200, 20)
    print(f"RMSNorm            M={M} K={K}: {us:6.2f} us  ({4 * M * K / us / 1e6:5.2f} TB/s)")
    for n_out in (0, round(0.01 * K)):
        ind = torch.randperm(K)[:n_out].to(torch.int32).to(dev) if n_out else None
        for bit in (8, 4):
            def fused():
                mixlib.RMSNormQuantFused(x[i[0] % 32], w, out, 1e-5, ind, xs, bit, packed=True); i[0] += 1
            us = time_graph(fused, 200, 20)
            print(f"RMSNorm+quant bit{bit} M={M} K={K} n_out={n_out:3d}: {us:6.2f} us  ({(4 * M * K + M * K * bit / 8) / us / 1e6:5.2f} TB/s)")
