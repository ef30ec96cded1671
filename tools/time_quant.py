#!/usr/bin/env python3
"""Time the fused extract+quantise kernel alone (graph replay, rotating pristine inputs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import mixlib
from tools.sweep_gemm import time_graph

M, K = 512, 4096
dev = "cuda"
for n_out in (0, 41):
    for packed in (False, True):
        for bit in (8, 4):
            x = torch.randn(64, M, K, device=dev).half()
            ind = torch.randperm(K)[:n_out].to(torch.int32).to(dev) if n_out else None
            xs = torch.zeros(M, 1, dtype=torch.float16, device=dev)
            i = [0]
            def f():
                mixlib.QuantFused(x[i[0] % 64], ind, xs, bit, 6.0, packed=packed)
                i[0] += 1
            us = time_graph(f, 200, 20)
            print(f"QuantFused M={M} K={K} bit={bit} n_out={n_out} packed={packed}: {us:.2f} us  ({(M*K*2 + M*K*bit/8)/us/1e6:.2f} TB/s)")
