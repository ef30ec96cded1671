#!/usr/bin/env python3
"""4-bit layer forward at small batches: FP6-coded weights ALONE (the default: one resident image), FP6 + the opt-in nibble image for batches
<= 32 rows (MixqConfig.small_batch_m4 = 32), and nibble weights only (MixqConfig.pack_fmt4 = FMT_P16X64: skinny / LDS-staged kernels); us per forward in a graph of
50 forwards, frozen layer."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import MixLibCache, MixLinear_GEMM, MixqConfig, _capi
from mixq_amd._capi import FMT_F6X128, FMT_P16X64
dev = "cuda"
names = _capi.gemm_config_names()
for (K, N) in [(4096, 11008), (4096, 4096), (11008, 4096)]:
    for M in (1, 16, 32, 64, 128, 256):
        row = []
        for fmt, small in ((FMT_F6X128, 0), (FMT_F6X128, 32), (FMT_P16X64, 0)):
            torch.manual_seed(0)
            cache = MixLibCache(M, bit=4, device=dev, config=MixqConfig(pack_fmt4=fmt, small_batch_m4=small))
            ls = torch.ones(K); ls[torch.randperm(K)[:128]] = 20.0
            layer = MixLinear_GEMM.from_linear(torch.nn.Linear(K, N, bias=False).half(), 4, cache=cache, layer_scales=ls, dev=dev)
            x = torch.randn(8, M, K).half().to(dev)
            for i in range(3):
                layer(x[i].clone(), None, True)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for i in range(50):
                        layer(x[i % 8], None, True)
                torch.cuda.synchronize()
                g.replay(); torch.cuda.synchronize()
                t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
                us = (time.perf_counter() - t0) * 1e6 / 50
            used = getattr(layer._packed_weight(M), "_mixq_fmt", 0)                   # (the FP6 layer serves small batches from its nibble image)
            cfg = names[_capi.load().mixq_gemm_pick_config_fmt(M, N, K, 4, used)] + (" on its nibble image" if used != fmt else "")
            row.append(f"{('fp6+nibble' if small else 'fp6 only') if fmt == FMT_F6X128 else 'nibble only'} {us:6.1f} us ({cfg})")
        print(f"{K:6d}->{N:6d} M={M:4d}: " + "   ".join(row), flush=True)
