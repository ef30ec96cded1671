"""A `mixq_amd.mixlib`-shaped backend on top of the CPU oracle, for testing the operator's HOST logic (state machine,
buffer management) on machines without a GPU.  Test infrastructure: installed with tests/conftest.py's swap_backend(mixq_amd.linear, ...) by
the tests only; the product never imports it."""
import numpy as np
import torch

from oracle import oracle as O

ACT_NONE, ACT_SILU, ACT_SILU_MUL = 0, 1, 2
calls = []


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def FindRowScale(x, x_scale, M, K, bit=8):
    calls.append("FindRowScale")
    q, s = O.find_row_scale(_np(x.reshape(-1, K)[:M]), bit)
    x_scale.reshape(-1)[0:M] = torch.from_numpy(s)
    return torch.from_numpy(q)


def ExtractOutliersAndSetToZeros(ind, x):
    calls.append("ExtractOutliersAndSetToZeros")
    xn = _np(x).copy()
    out = O.extract_outliers_zero(xn, _np(ind))
    x.copy_(torch.from_numpy(xn))
    return torch.from_numpy(out)


def QuantFused(x, ind, x_scale, bit, sigma, x_out=None, flag=None, n_dev=None, packed=False):
    calls.append("QuantFused")
    assert not packed
    M, K = x.shape
    xo = None
    if ind is not None and ind.numel():
        xo = ExtractOutliersAndSetToZeros(ind, x)
        calls.pop()
    q, s = O.find_row_scale(_np(x), bit)
    x_scale.reshape(-1)[0:M] = torch.from_numpy(s)
    if flag is not None and O.mispredicted(s, sigma, bit):
        flag |= 1
    return torch.from_numpy(q), xo


def DetectOutlierCols(x, sigma, scratch=None):
    calls.append("DetectOutlierCols")
    ind = O.find_outliers(_np(x), sigma)
    buf = torch.zeros(x.shape[1], dtype=torch.int32)
    buf[: ind.size] = torch.from_numpy(ind)
    return buf, torch.tensor([ind.size], dtype=torch.int32)


def DequantWeightCols(q_w, scale_col, ind, bit, out=None):
    calls.append("DequantWeightCols")
    return torch.from_numpy(O.dequant_weight_cols(_np(q_w), _np(scale_col), _np(ind), bit))


def unpack_int4_to_fp16(q_w, ind):
    calls.append("unpack_int4_to_fp16")
    return torch.from_numpy(O.unpack_i4_cols(_np(q_w), _np(ind)))


def FusedLinear(q_x, q_w, x_scale, scale_col, x_out, w_out, n_out, bias, M, N, K, bit=8, act=ACT_NONE, n_out_dev=None,
                addend=None, out=None, x_packed=False, w_packed=False):
    calls.append("FusedLinear")
    assert not x_packed and not w_packed
    xo = _np(x_out) if (n_out and x_out is not None) else None
    wo = _np(w_out) if (n_out and w_out is not None) else None
    y = O.linear_fused(_np(q_x), _np(q_w), _np(x_scale.reshape(-1)[0:M]), _np(scale_col), xo=xo, wo=wo,
                       addend=None if addend is None else _np(addend), bias=None if bias is None else _np(bias), act=act, bit=bit)
    return torch.from_numpy(y)


def layernorm_forward_cuda(x, weight, out, eps):
    calls.append("layernorm_forward_cuda")
    K = x.shape[-1]
    y = O.rmsnorm(_np(x.reshape(-1, K)), _np(weight), eps)
    out.reshape(-1, K).copy_(torch.from_numpy(y))
    return out


def RMSNormQuantFused(x, weight, out, eps, ind, x_scale, bit, sigma=6.0, flag=None, packed=False):
    calls.append("RMSNormQuantFused")
    assert not packed
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    M = x2.shape[0]
    idx = np.zeros(0, np.int32) if ind is None else _np(ind)
    y, xo, q, s = O.rmsnorm_quant(_np(x2), _np(weight), eps, idx, bit)
    out.reshape(-1, K).copy_(torch.from_numpy(y))
    x_scale.reshape(-1)[0:M] = torch.from_numpy(s)
    if flag is not None and O.mispredicted(s, sigma, bit):
        flag |= 1
    return torch.from_numpy(q), (torch.from_numpy(xo) if idx.size else None)


def PackW8A16(q_weight_kn):
    calls.append("PackW8A16")
    return q_weight_kn                        # the oracle reads the checkpoint layout directly


def W8A16Linear(x, w_packed, scale_col, bias, N, K, out=None):
    calls.append("W8A16Linear")
    y = O.w8a16_linear(_np(x.reshape(-1, K)), _np(w_packed), _np(scale_col), None if bias is None else _np(bias))
    return torch.from_numpy(y)
