#!/usr/bin/env python3
"""int4 quantiser (extract + scale + quantise) per output format: plain nibbles, P16X64, F6X128 (FP6 codes); us per launch in a graph."""
import os, sys, torch
os.environ.setdefault("MIXQ_TUNING_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib
from tools.sweep_gemm import time_graph
dev = "cuda"; lib = _capi.load()
for (M, K, n_out) in [(512, 4096, 128), (512, 11008, 128)]:
    nb = 32
    x = torch.randn(nb, M, K, device=dev).half()
    ind = torch.randperm(K)[:n_out].to(torch.int32).to(dev)
    for fmt, probe in ((0, 0), (1, 0), (3, 0), (4, 0), (3, 16)):
        row = []
        for cfg in (-1, 6, 7, 8, 9):
            assert lib.mixq_quant_set_config(cfg) == 0
            if probe:
                assert lib.mixq_quant_set_config(100 + probe) == 0
            xs = torch.zeros(M, 1, dtype=torch.float16, device=dev); i = [0]
            def f():
                mixlib.QuantFused(x[i[0] % nb], ind, xs, 4, 6.0, fmt=fmt); i[0] += 1
            row.append(time_graph(f, 200, 20))
        lib.mixq_quant_set_config(100); lib.mixq_quant_set_config(-1)
        print(f"QuantFused bit=4 M={M} K={K} n_out={n_out} fmt={fmt}{' (probe: row-contiguous stores)' if probe else ''}: auto={row[0]:.2f} 256x1={row[1]:.2f} 256x2={row[2]:.2f} 512x1={row[3]:.2f} 512x2={row[4]:.2f}", flush=True)
