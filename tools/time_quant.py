#!/usr/bin/env python3
"""Time the fused extract+quantise kernel alone (graph replay, rotating pristine inputs) for every launch geometry
(mixq_quant_set_config) and output format.  HBM bytes per launch: 2 M K read + M K bit/8 written (+ outlier columns)."""
import os, sys
import torch
os.environ.setdefault("MIXQ_TUNING_LIB", "1")      # ablation kernels / trace stamps / probe knobs live in the tools build (make tuning)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib
from tools.sweep_gemm import time_graph

dev = "cuda"
lib = _capi.load()
NAMES = ["r1 256x1", "64x1", "64x2", "64x4", "128x1", "128x2", "256x1", "256x2", "512x1", "512x2"]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
PROBE = "--probe" in sys.argv                   # timing probes of the (256,1) geometry: what the outlier handling costs, part by part
COLD = "--cold" in sys.argv                       # rotate through ~1.7 GB of inputs (beyond the 256 MB MALL), as bench.py's steps do
shapes = [(512, 4096)] if not args else [tuple(int(v) for v in a.split("x")) for a in args]
for (M, K) in shapes:
    nb = max(2, min(400 if COLD else 64, ((1700 << 20) if COLD else (1 << 30)) // (M * K * 2)))
    x = torch.randn(nb, M, K, device=dev).half()
    for n_out in (0, round(0.01 * K)):
        ind = torch.randperm(K)[:n_out].to(torch.int32).to(dev) if n_out else None
        for fmt in ((1,) if COLD else (0, 1, 2)):
            for bit in ((8,) if COLD else (8, 4)):
                row = []
                for cfg in range(len(NAMES)):
                    assert lib.mixq_quant_set_config(cfg) == 0
                    xs = torch.zeros(M, 1, dtype=torch.float16, device=dev)
                    i = [0]
                    def f():
                        mixlib.QuantFused(x[i[0] % nb], ind, xs, bit, 6.0, fmt=fmt)
                        i[0] += 1
                    us = time_graph(f, 400 if COLD else 200, 400 if COLD else 20)
                    row.append(us)
                lib.mixq_quant_set_config(-1)
                best = min(range(len(row)), key=lambda c: row[c])
                gbs = (M * K * 2 + M * K * bit / 8) / row[best] / 1e6
                print(f"QuantFused M={M} K={K} bit={bit} n_out={n_out} fmt={fmt}: " + "  ".join(f"{NAMES[c]}={row[c]:.2f}" for c in range(len(row))) +
                      f"  us | best {NAMES[best]} {row[best]:.2f} us = {gbs:.2f} TB/s", flush=True)

if PROBE:
    M, K = 512, 4096
    x = torch.randn(64, M, K, device=dev).half()
    ind = torch.randperm(K)[:41].to(torch.int32).to(dev)
    xs = torch.zeros(M, 1, dtype=torch.float16, device=dev)
    lib.mixq_quant_set_config(6)
    for dbg, what in [(0, "full"), (1, "no in-place zeroing of x"), (2, "no gather loads"), (8, "no x_out stores"), (1 | 2 | 8, "mask only"), (4, "no outlier handling at all")]:
        lib.mixq_quant_set_config(100 + dbg)
        i = [0]
        def f():
            mixlib.QuantFused(x[i[0] % 64], ind, xs, 8, 6.0, fmt=1)
            i[0] += 1
        print(f"probe {what:32s} {time_graph(f, 200, 20):.2f} us", flush=True)
    lib.mixq_quant_set_config(100)
    lib.mixq_quant_set_config(-1)
