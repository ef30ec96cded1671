"""numpy front-end of the CPU oracle (oracle/mixq_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
mixq_amd never does.  Parity status: see the header of mixq_oracle.c ("partially pinned": Python-level reference
arithmetic pinned by tests/golden/*, the native `mixlib` kernels unpinned because their source is not in
/root/reference).

All half-precision data crosses this API as numpy float16 arrays; integers as int8/uint8/int32.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmixq_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mixq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_find_outliers.restype = C.c_int
        _lib.orc_mispredicted.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _h(a):
    a = np.ascontiguousarray(a, dtype=np.float16)
    return a


def pack_i4(x):
    """linear.py:12-18."""
    x = np.ascontiguousarray(x, dtype=np.int8)
    R, K = x.shape
    out = np.empty((R, K // 2), dtype=np.uint8)
    lib().orc_pack_i4(_p(x), _p(out), R, K)
    return out


def unpack_i4_cols(w, ind):
    """linear.py:20-22 (mixlib.unpack_int4_to_fp16)."""
    w = np.ascontiguousarray(w, dtype=np.uint8)
    ind = np.ascontiguousarray(ind, dtype=np.int32)
    N, Kh = w.shape
    out = np.empty((N, ind.size), dtype=np.float16)
    lib().orc_unpack_i4_cols(_p(w), _p(ind), int(ind.size), _p(out), N, 2 * Kh)
    return out


def quant_weight_w8(w):
    """from_linear bit=8 (linear.py:111-119): returns (q_weight int8 [N,K], scale_col fp16 [1,N])."""
    w = _h(w)
    N, K = w.shape
    q = np.empty((N, K), dtype=np.int8)
    s = np.empty((1, N), dtype=np.float16)
    lib().orc_quant_weight_w8(_p(w), N, K, _p(q), _p(s))
    return q, s


def quant_weight_w4(w, ind):
    """from_linear bit=4 (linear.py:123-143) for a given fp-column set `ind`:
    returns (q_weight uint8 [N,K/2], scale_col fp16 [1,N], weight_cache fp16 [N,len(ind)])."""
    w = _h(w)
    ind = np.ascontiguousarray(ind, dtype=np.int32)
    N, K = w.shape
    q = np.empty((N, K // 2), dtype=np.uint8)
    s = np.empty((1, N), dtype=np.float16)
    wc = np.empty((N, ind.size), dtype=np.float16)
    lib().orc_quant_weight_w4(_p(w), N, K, _p(ind), int(ind.size), _p(q), _p(s), _p(wc))
    return q, s, wc


def find_outliers(x, sigma):
    """FindOutliers (linear.py:157-161) -> sorted distinct int32 column ids."""
    x = _h(x)
    M, K = x.shape
    out = np.empty(K, dtype=np.int32)
    n = lib().orc_find_outliers(_p(x), M, K, K, C.c_float(sigma), _p(out))
    return out[:n].copy()


def extract_outliers_zero(x, ind, ldo=None):
    """mixlib.ExtractOutliersAndSetToZeros: mutates x (a contiguous float16 array) in place, returns x_out [M,ldo]."""
    assert x.dtype == np.float16 and x.flags.c_contiguous
    ind = np.ascontiguousarray(ind, dtype=np.int32)
    M, K = x.shape
    n = int(ind.size)
    ldo = n if ldo is None else ldo
    out = np.empty((M, ldo), dtype=np.float16)
    lib().orc_extract_outliers_zero(_p(x), _p(ind), n, _p(out), M, K, K, ldo)
    return out


def find_row_scale(x, bit):
    """mixlib.FindRowScale -> (q, x_scale fp16 [M])."""
    x = _h(x)
    M, K = x.shape
    s = np.empty(M, dtype=np.float16)
    q = np.empty((M, K) if bit == 8 else (M, K // 2), dtype=np.int8 if bit == 8 else np.uint8)
    lib().orc_find_row_scale(_p(x), _p(s), _p(q), M, K, K, bit)
    return q, s


def mispredicted(x_scale, sigma, bit):
    """The predicate of linear.py:201."""
    s = _h(x_scale).reshape(-1)
    return bool(lib().orc_mispredicted(_p(s), int(s.size), C.c_float(sigma), bit))


def dequant_weight_cols(w, scale_col, ind, bit):
    """linear.py:207 / :209-210."""
    w = np.ascontiguousarray(w)
    ind = np.ascontiguousarray(ind, dtype=np.int32)
    N = w.shape[0]
    K = w.shape[1] * (1 if bit == 8 else 2)
    sc = None if scale_col is None else _h(scale_col).reshape(-1)
    out = np.empty((N, ind.size), dtype=np.float16)
    lib().orc_dequant_weight_cols(_p(w), _p(sc), _p(ind), int(ind.size), _p(out), N, K, int(ind.size), bit)
    return out


def gemm_i8(qx, qw):
    """mixlib.gemm (linear.py:235): exact int32 [M,N]."""
    qx = np.ascontiguousarray(qx, dtype=np.int8)
    qw = np.ascontiguousarray(qw, dtype=np.int8)
    M, K = qx.shape
    N = qw.shape[0]
    y = np.empty((M, N), dtype=np.int32)
    lib().orc_gemm_i8(_p(qx), _p(qw), _p(y), M, N, K)
    return y


def linear_fused(qx, qw, sx, sw, xo=None, wo=None, addend=None, bias=None, act=0, bit=8):
    """The compute step of linear.py:244-285 / :320-373 -> fp16 [M,N]."""
    qx = np.ascontiguousarray(qx)
    qw = np.ascontiguousarray(qw)
    M = qx.shape[0]
    N = qw.shape[0]
    K = qx.shape[1] * (1 if bit == 8 else 2)
    sx = _h(sx).reshape(-1)
    sw = _h(sw).reshape(-1)
    n_out = 0
    if xo is not None and wo is not None and xo.shape[1] > 0:
        xo, wo = _h(xo), _h(wo)
        n_out = xo.shape[1]
        assert wo.shape[1] == n_out
    else:
        xo = wo = None
    ad = None if addend is None else _h(addend)
    b = None if bias is None else _h(bias).reshape(-1)
    y = np.empty((M, N), dtype=np.float16)
    lib().orc_linear_fused(_p(qx), _p(qw), _p(sx), _p(sw), _p(xo), n_out, _p(wo), n_out, n_out, _p(ad), N, _p(b), _p(y), N,
                           M, N, K, act, bit)
    return y


def pair_rows_interleave(up, gate):
    """The row order MIXQ_ACT_SILU_PAIR reads two layers' per-channel arrays in (include/mixq_hip.h): row 4g+0 = up[2g], 4g+1 = up[2g+1],
    4g+2 = gate[2g], 4g+3 = gate[2g+1].  Written out element by element: the definition the product's helper is held to."""
    up, gate = np.asarray(up), np.asarray(gate)
    assert up.shape == gate.shape and up.shape[0] % 2 == 0
    out = np.empty((2 * up.shape[0],) + up.shape[1:], dtype=up.dtype)
    for c in range(up.shape[0]):
        g, e = divmod(c, 2)
        out[4 * g + e] = up[c]
        out[4 * g + 2 + e] = gate[c]
    return out


def pair_rows_split(joint):
    joint = np.asarray(joint)
    n = joint.shape[0] // 2
    up = np.empty((n,) + joint.shape[1:], dtype=joint.dtype)
    gate = np.empty_like(up)
    for c in range(n):
        g, e = divmod(c, 2)
        up[c], gate[c] = joint[4 * g + e], joint[4 * g + 2 + e]
    return up, gate


def linear_fused_pair(qx, qw2, sx, sw2, xo=None, wo2=None, bias2=None, bit=8):
    """gate_proj + up_proj over INTERLEAVED operands -> fp16 [M, N/2]: what mixquant/modules/fused/mlp.py:57-63 computes from the two
    layers - up_output = up_proj(x) (linear.py:244-285, an fp16 tensor), gate_output = silu-fused gate_proj(x) (linear.py:320-373),
    gate_output *= up_output - restated as up_proj's compute step followed by gate_proj's with the multiplier folded in (act = 2)."""
    qw_u, qw_g = pair_rows_split(qw2)
    sw_u, sw_g = pair_rows_split(_h(sw2).reshape(-1))
    wo_u = wo_g = b_u = b_g = None
    if wo2 is not None and xo is not None and np.asarray(xo).shape[1] > 0:
        wo_u, wo_g = pair_rows_split(_h(wo2))
    if bias2 is not None:
        b_u, b_g = pair_rows_split(_h(bias2).reshape(-1))
    up = linear_fused(qx, qw_u, sx, sw_u, xo=xo, wo=wo_u, addend=None, bias=b_u, act=0, bit=bit)
    return linear_fused(qx, qw_g, sx, sw_g, xo=xo, wo=wo_g, addend=up, bias=b_g, act=2, bit=bit)


def linear_dequant_ref(qx, qw, sx, sw, xo=None, ind=None, bias=None, wo=None):
    """north_star's gate: CPU Linear over the same dequantised operands, fp64 -> float64 [M,N] (int8 operands).
    wo: the fp16 weight_cache the reference multiplies the outlier columns with (linear.py:207); None -> qw*sw."""
    qx = np.ascontiguousarray(qx, dtype=np.int8)
    qw = np.ascontiguousarray(qw, dtype=np.int8)
    M, K = qx.shape
    N = qw.shape[0]
    sx = _h(sx).reshape(-1)
    sw = _h(sw).reshape(-1)
    n_out = 0
    if xo is not None and ind is not None and len(ind):
        xo = _h(xo)
        ind = np.ascontiguousarray(ind, dtype=np.int32)
        n_out = int(ind.size)
    else:
        xo = ind = None
    b = None if bias is None else _h(bias).reshape(-1)
    if wo is not None and n_out:
        wo = _h(wo)
        assert wo.shape == (N, n_out)
    else:
        wo = None
    y = np.empty((M, N), dtype=np.float64)
    lib().orc_linear_dequant_ref(_p(qx), _p(qw), _p(sx), _p(sw), _p(xo), n_out, _p(ind), n_out, _p(wo), n_out, _p(b), _p(y),
                                 M, N, K)
    return y


def unpack_i4_all(packed):
    """All nibbles of a packed [R,K/2] uint8 matrix as int8 [R,K] (inverse of pack_i4)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    lo = (packed & 0xF).astype(np.int8)
    hi = (packed >> 4).astype(np.int8)
    lo = np.where(lo >= 8, lo - 16, lo)
    hi = np.where(hi >= 8, hi - 16, hi)
    out = np.empty((packed.shape[0], packed.shape[1] * 2), dtype=np.int8)
    out[:, 0::2] = lo
    out[:, 1::2] = hi
    return out


def rmsnorm(x, w, eps):
    """mixlib.layernorm_forward_cuda (norm.py:21) -> fp16 [M,K]."""
    x, w = _h(x), _h(w).reshape(-1)
    M, K = x.shape
    out = np.empty((M, K), dtype=np.float16)
    lib().orc_rmsnorm(_p(x), _p(w), _p(out), M, K, K, K, C.c_float(eps))
    return out


def rmsnorm_quant(x, w, eps, ind, bit):
    """mixlib.layernorm_forward_cuda_extract_outliers[_int4] (norm.py:24-33) -> (out, x_out [M,n], q, x_scale [M])."""
    x, w = _h(x), _h(w).reshape(-1)
    ind = np.ascontiguousarray(ind, dtype=np.int32)
    M, K = x.shape
    n = int(ind.size)
    out = np.empty((M, K), dtype=np.float16)
    xo = np.empty((M, n), dtype=np.float16)
    s = np.empty(M, dtype=np.float16)
    q = np.empty((M, K) if bit == 8 else (M, K // 2), dtype=np.int8 if bit == 8 else np.uint8)
    lib().orc_rmsnorm_quant(_p(x), _p(w), _p(out), _p(ind), n, _p(s), _p(q), _p(xo) if n else None, M, K, K, K, n,
                            C.c_float(eps), bit)
    return out, xo, q, s


def quant_weight_w8a16(wkn):
    """EETQ quant_weights(W^T [K,N], int8) as used at linear.py:102-106 -> (q int8 [K,N], scale fp16 [N]).  PARITY
    UNPINNED (EETQ absent): restates FasterTransformer's symmetric per-column quantisation (mixq_oracle.c)."""
    wkn = _h(wkn)
    K, N = wkn.shape
    q = np.empty((K, N), dtype=np.int8)
    s = np.empty(N, dtype=np.float16)
    lib().orc_quant_weight_w8a16(_p(wkn), K, N, _p(q), _p(s))
    return q, s


def w8a16_linear(x, q, scale, bias=None):
    """EETQ w8_a16_gemm(x, q_weight, scale) (+ bias, linear.py:178-184) -> fp16 [M,N]."""
    x = _h(x)
    q = np.ascontiguousarray(q, dtype=np.int8)
    scale = _h(scale).reshape(-1)
    M, K = x.shape
    N = q.shape[1]
    y = np.empty((M, N), dtype=np.float16)
    b = None if bias is None else _h(bias).reshape(-1)
    lib().orc_w8a16_linear(_p(x), K, _p(q), _p(scale), _p(b) if b is not None else None, _p(y), N, M, N, K)
    return y
