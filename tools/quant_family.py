#!/usr/bin/env python3
"""The memory-bound kernels of the path - quant_rows2_kernel (extract + quantise), rmsnorm_kernel (RMSNorm + extract + quantise),
quant_known_kernel (quantise with the row maxima the producing GEMM left) - alone, at K = 4096 and K = 11008, 512 token rows:
  default      us per launch by graph replay over rotating inputs, and algorithmic bytes / time against the 8 TB/s HBM peak
  --eager N    N eager launches of each (what the rocprofv3 --pmc passes of tools/r05_profile.sh run: counters need ungraphed dispatches)
Replaces /root/reference/mixquant/modules/linear.py:187-193 (FindRowScale + ExtractOutliersAndSetToZeros) and fused/norm.py:24-33."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import FasterTransformerRMSNorm, MixLibCache, MixLinear_GEMM, MixLlamaMLP, _capi, mixlib  # noqa: E402
from mixq_amd.linear import kept_outlier_map  # noqa: E402
from tools.sweep_gemm import time_graph  # noqa: E402

dev = "cuda"
M = 512
NB = 24                                                 # rotating inputs (beyond one XCD's L2 at either K)


def cases():
    """(label, kernel-name pattern, algorithmic bytes per launch, callable(i))"""
    out = []
    for K in (4096, 11008):
        n_out = round(0.01 * K)
        g = torch.Generator().manual_seed(K)
        ind = torch.randperm(K, generator=g)[:n_out].to(torch.int32).to(dev)
        x = torch.randn(NB, M, K, device=dev).half()
        x[:, :, ind.long()] *= 20
        xs = torch.zeros(M, 1, dtype=torch.float16, device=dev)
        ldo = mixlib._pad16(n_out)
        x_out = torch.zeros(M, ldo, dtype=torch.float16, device=dev)
        kept = kept_outlier_map(ind, K)
        for bit in (8, 4):
            alg = M * K * 2 + M * K * bit // 8 + M * ldo * 2 + M * 2
            def f(i, x=x, ind=ind, xs=xs, bit=bit, x_out=x_out):
                mixlib.QuantFused(x[i % NB], ind, xs, bit, 6.0, x_out=x_out, packed=True)
            out.append((f"quant_rows2   K={K:5d} W{bit} {n_out:3d} cols, mask built in the kernel", alg, f))
            if kept is not None:
                def fk(i, x=x, ind=ind, xs=xs, bit=bit, x_out=x_out, kept=kept):
                    mixlib.QuantFused(x[i % NB], ind, xs, bit, 6.0, x_out=x_out, packed=True, col_mask=kept)
                out.append((f"quant_rows2   K={K:5d} W{bit} {n_out:3d} cols, kept outlier map", alg, fk))
        # RMSNorm + extract + quantise: reads x and the weight, writes q, x_out, x_scale (the fp16 normalised row is not stored on this route)
        w = (torch.rand(K, device=dev) + 0.5).half()
        o = torch.empty(M, K, dtype=torch.float16, device=dev)
        for bit in (8,):
            alg = M * K * 2 + K * 2 + M * K * bit // 8 + M * ldo * 2 + M * 2
            def fn(i, x=x, w=w, o=o, ind=ind, xs=xs, bit=bit):
                mixlib.RMSNormQuantFused(x[i % NB], w, o, 1e-5, ind, xs, bit, packed=True)
            out.append((f"rmsnorm_quant K={K:5d} W{bit} {n_out:3d} cols", alg, fn))
    return out


def known_case():
    """quant_known_kernel as the MLP block runs it: down_proj's quantise pass with the row maxima gate_proj's epilogue left (K = 11008)."""
    H, F = 4096, 11008
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
    for _ in range(3):
        mlp(norm(base.clone()))
    torch.cuda.synchronize()
    xs = base.unsqueeze(0).repeat(NB, 1, 1).contiguous()
    def f(i):
        mlp(norm(xs[i % NB]))
    return f, (lambda: xs.copy_(base.unsqueeze(0).expand_as(xs)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--eager", type=int, default=0)
    ap.add_argument("--no-block", action="store_true")
    args = ap.parse_args()
    print(_capi.device_info())
    cs = cases()
    if args.eager:
        for label, alg, f in cs:
            for i in range(args.eager):
                f(i)
        if not args.no_block:
            f, restore = known_case()
            for i in range(args.eager):
                f(i)
        torch.cuda.synchronize()
        print("eager launches done")
        return
    print(f"us per launch (graph of 200 launches back to back, {NB} rotating inputs), algorithmic bytes / time, fraction of the 8 TB/s HBM peak; M = {M}")
    for label, alg, f in cs:
        i = [0]
        def g():
            f(i[0]); i[0] += 1
        us = time_graph(g, 200, 20)
        print(f"  {label:64s} {us:6.2f} us  {alg / 1e6:6.2f} MB  {alg / us / 1e6:5.2f} TB/s  {alg / us / 1e6 / 8.0:5.3f}", flush=True)


if __name__ == "__main__":
    main()
