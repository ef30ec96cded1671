#!/usr/bin/env python3
"""Timing of the route the UNCHANGED reference forward takes on gfx950 (torch reports capability major 9 ->
/root/reference/mixquant/modules/linear.py:234-241): y = mixlib.gemm(q, W, M, N, K); mm = torch.mm(X_out, weight_cache.T);
y1 = mixlib.dequantizeInt8(y, x_scale, scale_col, mm, 8, M, N) - with the deferred product (PendingGemmI32: one fused kernel
behind the two calls) and as the literal pair (int32 round trip + vectorised dequantisation), next to the fused entry point the
non-Hopper branch calls (int8FusedDequantize) and to the native operator.  One hipGraph of 50 sequences each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import MixLibCache, MixLinear_GEMM, _capi, mixlib
from tools.sweep_gemm import time_graph

dev = "cuda"
print(_capi.device_info(), "capability", torch.cuda.get_device_capability())
M, K, N = 512, 4096, 11008
torch.manual_seed(0)
lin = torch.nn.Linear(K, N, bias=False).half()
cache = MixLibCache(M, device=dev)
layer = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=dev)
cols = torch.randperm(K, generator=torch.Generator().manual_seed(1))[:41]
base = torch.randn(M, K).half(); base[:, cols] *= 20
base = base.to(dev)
q_weight, scale_col = layer.q_weight.clone(), layer.scale_col
for _ in range(3):
    layer(base.clone(), None, True)
ind, weight_cache = layer.ind, layer.weight_cache.contiguous()
xs = [base.clone() for _ in range(8)]
i = [0]
def seq_arch9():
    x = xs[i[0] % 8]; i[0] += 1
    cache.activation_outliers = mixlib.ExtractOutliersAndSetToZeros(ind, x)
    cache.q_xcache = mixlib.FindRowScale(x, cache.x_scale, M, K, 8)
    y = mixlib.gemm(cache.q_xcache, q_weight, M, N, K)
    mm = torch.mm(cache.activation_outliers, weight_cache.T)
    return mixlib.dequantizeInt8(y, cache.x_scale, scale_col, mm, 8, M, N)
def seq_fused_branch():
    x = xs[i[0] % 8]; i[0] += 1
    cache.activation_outliers = mixlib.ExtractOutliersAndSetToZeros(ind, x)
    cache.q_xcache = mixlib.FindRowScale(x, cache.x_scale, M, K, 8)
    mm = torch.mm(cache.activation_outliers, weight_cache.T)
    return mixlib.int8FusedDequantize(cache.q_xcache, q_weight, cache.x_scale, scale_col, mm, M, N, K)
def native():
    x = xs[i[0] % 8]; i[0] += 1
    return layer(x, None, True)
flops = 2.0 * M * N * K
for name, lazy, packed, fo, fn in [("arch==9 route, deferred product (default)", True, False, False, seq_arch9),
                                   ("arch==9 route, literal pair (ShimConfig.literal())", False, False, False, seq_arch9),
                                   ("fused branch (arch != 9): int8FusedDequantize, plain operands", True, False, False, seq_fused_branch),
                                   ("arch==9 route, deferred product + packed_operands", True, True, False, seq_arch9),
                                   ("fused branch + packed_operands", True, True, False, seq_fused_branch),
                                   ("arch==9 route + fused_outliers", True, False, True, seq_arch9),
                                   ("arch==9 route + packed_operands + fused_outliers", True, True, True, seq_arch9),
                                   ("fused branch + packed_operands + fused_outliers", True, True, True, seq_fused_branch),
                                   ("arch==9 route + all three switches (packed operands, fused outliers, fused prepass)", True, True, 2, seq_arch9),
                                   ("fused branch + all three switches", True, True, 2, seq_fused_branch),
                                   ("native operator (packed operands, fused quantise + GEMM)", True, False, False, native)]:
    with mixlib.configured(mixlib.ShimConfig(lazy_gemm=lazy, packed_operands=packed, fused_outliers=bool(fo), fused_prepass=fo == 2)):
        us = time_graph(fn, 50, 10)
    print(f"{name:88s} {us:8.2f} us / forward  {flops / us / 1e6:7.1f} TFLOPS ({100 * flops / us / 1e6 / 5033:4.1f} % of peak)", flush=True)
