#!/usr/bin/env python3
"""First-contact GPU sanity check of every kernel against torch expressions on the same GPU (development aid;
the parity tests proper live in tests/ and compare with the CPU oracle)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import mixlib, MixLinear_GEMM, MixLibCache  # noqa: E402
from mixq_amd.linear import pack_to_i4  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
ok = True


def report(name, cond, extra=""):
    global ok
    ok &= bool(cond)
    print(("PASS " if cond else "FAIL ") + name + " " + extra, flush=True)


def ref_quant(x, bit):
    qmax = 2 ** (bit - 1) - 1
    amax = x.float().abs().amax(dim=1, keepdim=True)
    s = (amax / qmax).half()
    sf = s.float()
    q = torch.where(sf > 0, torch.round(x.float() / sf), torch.zeros_like(sf)).clamp(-qmax, qmax)
    return q.to(torch.int8), s


for (M, K) in [(512, 4096), (37, 11008), (5, 64), (512, 28672)]:
    x = torch.randn(M, K, device=dev).half()
    x[0] = 0
    for bit in (8, 4):
        xs = torch.zeros(M, 1, dtype=torch.float16, device=dev)
        q = mixlib.FindRowScale(x, xs, M, K, bit)
        qr, sr = ref_quant(x, bit)
        if bit == 4:
            qr = pack_to_i4(qr.cpu()).to(dev)
        report(f"FindRowScale M={M} K={K} bit={bit}", torch.equal(xs, sr) and torch.equal(q, qr),
               f"scale_eq={torch.equal(xs, sr)} q_mismatch={(q != qr).sum().item()}")

# fused extract + quant + flag + detect
M, K = 512, 4096
x = torch.randn(M, K, device=dev).half()
cols = torch.randperm(K)[:41].sort()[0].to(torch.int32).to(dev)
x[:, cols.long()] *= 20
x0 = x.clone()
ind_buf, count = mixlib.DetectOutlierCols(x, 6.0)
n = int(count.item())
ref_ind = torch.unique(torch.where(x0.abs() > 6.0)[1]).to(torch.int32)
report("DetectOutlierCols", n == ref_ind.numel() and torch.equal(ind_buf[:n], ref_ind), f"n={n} ref={ref_ind.numel()}")
ind = ref_ind
xs = torch.zeros(M, 1, dtype=torch.float16, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
q, xo = mixlib.QuantFused(x, ind, xs, 8, 6.0, flag=flag)
xz = x0.clone()
xz[:, ind.long()] = 0
qr, sr = ref_quant(xz, 8)
report("QuantFused q/scale", torch.equal(q, qr) and torch.equal(xs, sr))
report("QuantFused x_out", torch.equal(xo, x0[:, ind.long()]))
report("QuantFused zeroed x in place", torch.equal(x, xz))
report("QuantFused flag", int(flag.item()) == int((sr.float().max() > torch.tensor(6.0).half().float() / 127).item()), f"flag={flag.item()}")
x2 = x0.clone()
xo2 = mixlib.ExtractOutliersAndSetToZeros(ind, x2)
report("ExtractOutliersAndSetToZeros", torch.equal(xo2, x0[:, ind.long()]) and torch.equal(x2, xz))

# weights
N = 1024
W = (torch.rand(N, K, device=dev) * 2 - 1) / 64
lin = torch.nn.Linear(K, N, bias=True).to(dev).half()
cache = MixLibCache(512, device=dev)
ql = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=dev)
wc = mixlib.DequantWeightCols(ql.q_weight, ql.scale_col, ind, 8)
wc_ref = ql.q_weight[:, ind.long()].to(torch.float16) * ql.scale_col.T
report("DequantWeightCols int8", torch.equal(wc, wc_ref))

# full operator, 3 calls (outliers appear on call 1), vs Linear over dequantised operands
for call in range(3):
    xin = x0.clone()
    y = ql(xin, None, True)
    Wd = ql.q_weight.double() * ql.scale_col.double().T
    Xd = qr.double() * sr.double()
    Xd[:, ind.long()] = x0[:, ind.long()].double()
    yref = Xd @ Wd.T + lin.bias.double()
    err = (y.double() - yref).abs().max().item()
    report(f"MixLinear_GEMM forward call {call}", err < 1e-2, f"max_abs_err={err:.3e} n_ind={ql.ind.numel()} add={ql.add_outliers}")
yfp = torch.nn.functional.linear(x0, lin.weight, lin.bias)
print("info: err vs unquantised fp16 linear:", (y.float() - yfp.float()).abs().max().item())

# reference-style flow through the mixlib surface: torch.mm addend path
q8, _ = ref_quant(xz, 8)
addend = torch.mm(x0[:, ind.long()], wc_ref.T)
y2 = mixlib.int8FusedDequantize(q8, ql.q_weight, sr, ql.scale_col, addend, M, N, K) + lin.bias
report("int8FusedDequantize + addend", (y2.double() - yref).abs().max().item() < 1e-2, f"{(y2.double() - yref).abs().max().item():.3e}")
y3 = mixlib.int8FusedDequantize(q8, ql.q_weight, sr, ql.scale_col, cache.zeros, M, N, K)
y32 = mixlib.gemm(q8, ql.q_weight, M, N, K)
i32ref = (q8.double() @ ql.q_weight.double().T)
report("gemm int32 exact", torch.equal(y32.double(), i32ref))
y4 = mixlib.dequantizeInt8(y32, sr, ql.scale_col, cache.zeros, 8, M, N)
report("dequantizeInt8 == fused", (y3.float() - y4.float()).abs().max().item() <= 2e-3, f"{(y3.float() - y4.float()).abs().max().item():.3e}")
ys = mixlib.int8FusedDequantizeSilu(q8, ql.q_weight, sr, ql.scale_col, cache.zeros, M, N, K)
report("Silu epilogue", (ys.float() - torch.nn.functional.silu(y3.float())).abs().max().item() < 5e-3)

# W4A4
scales = torch.rand(K)
q4 = MixLinear_GEMM.from_linear(lin, 4, cache=cache, layer_scales=scales, dev=dev)
x4 = torch.randn(M, K, device=dev).half()
x4ref = x4.clone()
y = q4(x4, None, True)
ind4 = q4.ind.long()
xz4 = x4ref.clone(); xz4[:, ind4] = 0
qr4, sr4 = ref_quant(xz4, 4)
u = mixlib.unpack_int4_to_fp16(q4.q_weight, torch.arange(K, dtype=torch.int32, device=dev))
Wd = u.double() * q4.scale_col.double().T
yref = (qr4.double() * sr4.double()) @ Wd.T + x4ref[:, ind4].double() @ q4.weight_cache.double().T + lin.bias.double()
err = (y.double() - yref).abs().max().item()
report("W4A4 forward", err < 1e-2, f"max_abs_err={err:.3e}")
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
