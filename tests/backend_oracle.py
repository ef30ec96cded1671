"""A `mixq_amd.mixlib`-shaped backend on top of the CPU oracle, for testing the operator's HOST logic (state machine, derived buffers,
packed images, kept argument blocks, the joint gate / up image) on machines without a GPU.  Test infrastructure: installed with
tests/conftest.py's swap_backend(mixq_amd.linear / fused / eetq, ...) by the tests only; the product never imports it.

Round 6: the FULL surface the product modules call - packed operand formats (real byte layouts of include/mixq_hip.h, so that the product's
own host-side unpacking, state_dict of a compacted layer, the joint image's row split ... run on them), capacity-padded `ind` + device
count, kept outlier maps, ForwardPlan, the row-maximum side output, ACT_SILU_PAIR - so that mixq_amd.linear / fused have ONE code shape
(no `hasattr(_backend, ...)` branches): what runs here on the CPU is the code that runs on the GPU, with the kernels swapped for the
oracle.  Tensors live on the CPU; `dev_check` accepts them (the HIP backend's raises: the product has no CPU path)."""
import threading

import numpy as np
import torch

from oracle import oracle as O

ACT_NONE, ACT_SILU, ACT_SILU_MUL, ACT_SILU_PAIR = 0, 1, 2, 3
FMT_PLAIN, FMT_P16X64, FMT_F16X64, FMT_F6X128, FMT_R6X128 = 0, 1, 2, 3, 4
PAIR_LAUNCH = True
calls = []


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def dev_check(*ts):
    """(the HIP backend refuses CPU tensors here; this one takes them)"""
    for t in ts:
        if t is not None and t.is_cuda:
            raise RuntimeError("backend_oracle: a CPU stand-in; got a GPU tensor")


def set_fmt(t, fmt):
    t._mixq_fmt = fmt
    return t


def fmt_of(t):
    return getattr(t, "_mixq_fmt", FMT_PLAIN)


def packed_rows(rows):
    return (rows + 15) // 16 * 16


def kept_map_words(K):
    return (((K + 31) // 32 + 1 + 3) // 4) * 4 + ((K + 7) // 8) * 4


# ---- the packed layouts of include/mixq_hip.h, vectorised (tests/test_pack_properties.py holds the byte-by-byte restatements they are checked against)
_F6_MAG = np.array([0x00, 0x0C, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18], dtype=np.uint8)
_F6_OF_NIBBLE = np.array([(_F6_MAG[v] if v < 8 else (0x20 | _F6_MAG[16 - v])) for v in range(16)], dtype=np.uint8)
_NIBBLE_OF_F6 = np.zeros(64, dtype=np.uint8)
for _v in range(16):
    _NIBBLE_OF_F6[_F6_OF_NIBBLE[_v]] = _v
_NIBBLE_OF_F6[0x20] = 0                                   # (-0)
_SW = (-(np.arange(16) >> 2)) & 3                          # P16X64: chunk c of row r sits at c ^ _SW[r]
_P16_IDX = (np.arange(4)[None, :] ^ _SW[:, None]).reshape(1, 16, 1, 4, 1)


def _padded(q, rows16):
    a = np.zeros((rows16, q.shape[1]), dtype=np.uint8)
    a[: q.shape[0]] = q.view(np.uint8)
    return a


def _f6_fragments(q):
    """nibble-packed [R, K/2] -> FP6 code bytes [rb, r, kb, g, 24]: lane (g, r) of block (kb, rb) holds elements 32 g .. + 31 of row 16 rb + r
    as a little-endian stream of 6-bit codes."""
    R, KB = q.shape
    K = 2 * KB
    r16 = packed_rows(R)
    a = _padded(q, r16)
    nib = np.empty((r16, K), dtype=np.uint8)
    nib[:, 0::2], nib[:, 1::2] = a & 15, a >> 4
    codes = _F6_OF_NIBBLE[nib].astype(np.uint32).reshape(r16 // 16, 16, K // 128, 4, 8, 4)
    v = codes[..., 0] | (codes[..., 1] << 6) | (codes[..., 2] << 12) | (codes[..., 3] << 18)
    return np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], axis=-1).astype(np.uint8).reshape(r16 // 16, 16, K // 128, 4, 24)


def pack_np(q, fmt):
    q = np.ascontiguousarray(q).view(np.uint8)
    R, KB = q.shape
    r16 = packed_rows(R)
    if fmt in (FMT_P16X64, FMT_F16X64):
        assert KB % 64 == 0
        t = _padded(q, r16).reshape(r16 // 16, 16, KB // 64, 4, 16)           # [rb, r, kb, c, b]
        if fmt == FMT_P16X64:
            t = np.take_along_axis(t, np.broadcast_to(_P16_IDX, t.shape[:4] + (1,)), axis=3).transpose(2, 0, 1, 3, 4)
        else:
            t = t.transpose(2, 0, 3, 1, 4)                                       # [kb, rb, c, r, b]
        return np.ascontiguousarray(t).reshape(r16, KB)
    assert fmt in (FMT_F6X128, FMT_R6X128) and KB % 64 == 0
    f = _f6_fragments(q)                                                          # [rb, r, kb, g, 24]
    if fmt == FMT_F6X128:
        a16 = f[..., :16].transpose(2, 0, 3, 1, 4).reshape(KB // 64, r16 // 16, 1024)   # byte 16 (16 g + r) + b
        a8 = f[..., 16:].transpose(2, 0, 3, 1, 4).reshape(KB // 64, r16 // 16, 512)     # 1024 + 8 (16 g + r) + b
    else:
        a16 = f[..., :16].transpose(2, 0, 1, 3, 4).reshape(KB // 64, r16 // 16, 16, 64)  # row r: 16-byte pieces g at 96 r + 16 g
        a8 = f[..., 16:].transpose(2, 0, 1, 3, 4).reshape(KB // 64, r16 // 16, 16, 32)   # ... its 8-byte pieces at 96 r + 64 + 8 g
        return np.ascontiguousarray(np.concatenate([a16, a8], axis=-1)).reshape(r16, KB * 3 // 2)
    return np.ascontiguousarray(np.concatenate([a16, a8], axis=-1)).reshape(r16, KB * 3 // 2)


def unpack_np(p, R, fmt):
    p = np.ascontiguousarray(p).view(np.uint8)
    r16 = p.shape[0]
    if fmt in (FMT_P16X64, FMT_F16X64):
        KB = p.shape[1]
        b = p.reshape(KB // 64, r16 // 16, 1024)
        if fmt == FMT_P16X64:
            t = b.reshape(KB // 64, r16 // 16, 16, 4, 16).transpose(1, 2, 0, 3, 4)   # [rb, r, kb, p, b]
            t = np.take_along_axis(t, np.broadcast_to(_P16_IDX, t.shape[:4] + (1,)), axis=3)
        else:
            t = b.reshape(KB // 64, r16 // 16, 4, 16, 16).transpose(1, 3, 0, 2, 4)
        return np.ascontiguousarray(t).reshape(r16, KB)[:R].copy()
    KB = p.shape[1] * 2 // 3
    K = 2 * KB
    b = p.reshape(KB // 64, r16 // 16, 1536)
    if fmt == FMT_F6X128:
        f = np.concatenate([b[..., :1024].reshape(KB // 64, r16 // 16, 4, 16, 16), b[..., 1024:].reshape(KB // 64, r16 // 16, 4, 16, 8)], axis=-1)
        f = f.transpose(1, 3, 0, 2, 4)                                            # [rb, r, kb, g, 24]
    else:
        rows = b.reshape(KB // 64, r16 // 16, 16, 96)
        f = np.concatenate([rows[..., :64].reshape(KB // 64, r16 // 16, 16, 4, 16), rows[..., 64:].reshape(KB // 64, r16 // 16, 16, 4, 8)], axis=-1)
        f = f.transpose(1, 2, 0, 3, 4)
    tri = f.astype(np.uint32).reshape(r16 // 16, 16, K // 128, 4, 8, 3)
    v = tri[..., 0] | (tri[..., 1] << 8) | (tri[..., 2] << 16)
    codes = np.stack([(v >> (6 * i)) & 63 for i in range(4)], axis=-1).reshape(r16, K)
    nib = _NIBBLE_OF_F6[codes]
    return (nib[:, 0::2] | (nib[:, 1::2] << 4))[:R].copy()


def PackOperand(q, fmt=FMT_P16X64):
    calls.append("PackOperand")
    if q.dim() != 2 or not q.is_contiguous() or q.element_size() != 1:
        raise RuntimeError("PackOperand: expected a contiguous 2-D int8/uint8 tensor")
    out = torch.from_numpy(pack_np(_np(q), fmt))
    return set_fmt(out.view(q.dtype), fmt)


def UnpackOperand(packed, R, fmt=None):
    calls.append("UnpackOperand")
    fmt = fmt_of(packed) if fmt is None else fmt
    if fmt not in (FMT_P16X64, FMT_F16X64, FMT_F6X128, FMT_R6X128):
        raise RuntimeError("UnpackOperand: the tensor carries no packed-format tag; pass fmt")
    return torch.from_numpy(unpack_np(_np(packed), R, fmt)).view(packed.dtype)


def _plain(t, rows):
    """plain [rows, KB] numpy view of a possibly packed operand"""
    f = fmt_of(t)
    a = _np(t)
    return unpack_np(a, rows, f).view(a.dtype) if f else a[:rows]


def _as_fmt(q_np, fmt, dtype):
    if not fmt:
        return torch.from_numpy(q_np)
    return set_fmt(torch.from_numpy(pack_np(q_np, fmt)).view(dtype), fmt)


def _want_fmt(packed, fmt):
    return fmt if fmt is not None else (FMT_P16X64 if packed else FMT_PLAIN)


# ---- reference surface ----------------------------------------------------------------------------------------------------------------
def FindRowScale(x, x_scale, M, K, bit=8):
    calls.append("FindRowScale")
    q, s = O.find_row_scale(_np(x.reshape(-1, K)[:M]), bit)
    x_scale.reshape(-1)[0:M] = torch.from_numpy(s)
    return torch.from_numpy(q)


def FindRowScalePacked(x, x_scale, M, K, bit=8, fmt=FMT_P16X64):
    calls.append("FindRowScalePacked")
    q, s = O.find_row_scale(_np(x.reshape(-1, K)[:M]), bit)
    x_scale.reshape(-1)[0:M] = torch.from_numpy(s)
    return _as_fmt(q, fmt, torch.int8 if bit == 8 else torch.uint8)


def ExtractOutliersAndSetToZeros(ind, x):
    calls.append("ExtractOutliersAndSetToZeros")
    xn = _np(x).copy()
    out = O.extract_outliers_zero(xn, _np(ind))
    x.copy_(torch.from_numpy(xn))
    return torch.from_numpy(out)


def _live(ind, n_dev):
    """(live column ids, capacity) of a capacity-padded `ind` buffer + device count"""
    if ind is None or ind.numel() == 0:
        return np.zeros(0, np.int32), 0
    cap = int(ind.numel())
    n = cap if n_dev is None else min(int(n_dev.reshape(-1)[0].item()), cap)
    return _np(ind).astype(np.int32)[:n], cap


def _check_kept_map(col_mask, ind_live, K, who):
    if col_mask is None:
        return
    if col_mask.element_size() != 4 or col_mask.numel() < kept_map_words(K) or not col_mask.is_contiguous():
        raise RuntimeError(f"{who}: col_mask must be the kept outlier map of K = {K} columns ({kept_map_words(K)} int32 words)")
    m = _np(col_mask).view(np.uint32)
    W = (K + 31) // 32
    bits = np.zeros(W * 32, dtype=bool)
    bits[ind_live] = True
    want = np.packbits(bits.reshape(W, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1)
    if int(m[W]) == ind_live.size:                                   # (a map built for another live count is ignored by the kernels)
        assert np.array_equal(m[:W], want), f"{who}: the kept outlier map does not mark the live `ind` columns"
        keep = m[((W + 1 + 3) // 4) * 4:].view(np.uint16)[:K]
        assert np.array_equal(keep == 0, bits[:K]) and ((keep == 0) | (keep == 0xffff)).all(), f"{who}: AND-masks of the kept outlier map"


def QuantFused(x, ind, x_scale, bit, sigma, x_out=None, flag=None, n_dev=None, packed=False, fmt=None, col_mask=None):
    calls.append("QuantFused")
    fmt = _want_fmt(packed, fmt)
    if x.dtype != torch.float16:
        raise RuntimeError("QuantFused: x must be float16")
    M, K = x.shape
    if x_scale.numel() < M:
        raise RuntimeError(f"QuantFused: x_scale holds {x_scale.numel()} rows, the batch has {M} (MixLibCache.inputdim too small)")
    live, cap = _live(ind, n_dev)
    _check_kept_map(col_mask, live, K, "QuantFused")
    xo = None
    if cap:
        xn = _np(x).copy()
        vals = O.extract_outliers_zero(xn, live)
        x.copy_(torch.from_numpy(xn))
        xo = torch.zeros((M, (cap + 15) // 16 * 16), dtype=torch.float16) if x_out is None else x_out
        xo[:, :live.size] = torch.from_numpy(vals)
        xo[:, live.size:] = 0
    q, s = O.find_row_scale(_np(x), bit)
    x_scale.reshape(-1)[0:M] = torch.from_numpy(s)
    if flag is not None and O.mispredicted(s, sigma, bit):
        flag |= 1
    return _as_fmt(q, fmt, torch.int8 if bit == 8 else torch.uint8), (xo[:, :cap] if cap else None)


def DetectOutlierCols(x, sigma, scratch=None):
    calls.append("DetectOutlierCols")
    ind = O.find_outliers(_np(x), sigma)
    buf = torch.zeros(x.shape[1], dtype=torch.int32)
    buf[: ind.size] = torch.from_numpy(ind)
    return buf, torch.tensor([ind.size], dtype=torch.int32)


def DequantWeightCols(q_w, scale_col, ind, bit, out=None):
    calls.append("DequantWeightCols")
    if fmt_of(q_w) != FMT_PLAIN:
        raise RuntimeError("DequantWeightCols: q_w must be the plain, contiguous [N,KB] matrix")
    return torch.from_numpy(O.dequant_weight_cols(_np(q_w), _np(scale_col), _np(ind), bit))


def unpack_int4_to_fp16(q_w, ind):
    calls.append("unpack_int4_to_fp16")
    return torch.from_numpy(O.unpack_i4_cols(_np(q_w), _np(ind)))


def amax_supported(M, N, K, x_fmt, w_fmt):
    return x_fmt == FMT_P16X64 and w_fmt == FMT_F16X64 and M > 0 and K % 64 == 0


def _is_zero_addend(addend):
    return addend is None or getattr(addend, "_mixq_all_zero", False)


def FusedLinear(q_x, q_w, x_scale, scale_col, x_out, w_out, n_out, bias, M, N, K, bit=8, act=ACT_NONE, n_out_dev=None,
                addend=None, out=None, x_packed=None, w_packed=None, row_amax=None, col_mask=None):
    calls.append("FusedLinear")
    if x_scale.numel() < M or scale_col.numel() < N:
        raise RuntimeError("FusedLinear: x_scale / scale_col are shorter than M / N")
    xf, wf = fmt_of(q_x), fmt_of(q_w)
    if (xf in (FMT_F6X128, FMT_R6X128) or wf in (FMT_F6X128, FMT_R6X128)) and (xf != FMT_R6X128 or wf != FMT_F6X128):
        raise RuntimeError("the FP6 form of the W4A4 GEMM takes activations in R6X128 and weights in F6X128")
    if xf == FMT_F16X64:
        raise RuntimeError("F16X64 is a weight format")
    qx, qw = _plain(q_x, M), _plain(q_w, N)
    n_live = n_out if n_out_dev is None else min(int(n_out_dev.reshape(-1)[0].item()), n_out)
    xo = wo = None
    if n_live and x_out is not None and w_out is not None:
        xo, wo = np.ascontiguousarray(_np(x_out)[:, :n_live]), np.ascontiguousarray(_np(w_out)[:, :n_live])
    sx, sw = _np(x_scale.reshape(-1)[0:M]), _np(scale_col).reshape(-1)[:N]
    b = None if bias is None else _np(bias)
    if act == ACT_SILU_PAIR:
        if N % 16 or not _is_zero_addend(addend):
            raise RuntimeError("FusedLinear: ACT_SILU_PAIR needs N % 16 == 0 and takes no addend")
        y = O.linear_fused_pair(qx, qw, sx, sw, xo=xo, wo2=wo, bias2=b, bit=bit)
    else:
        if act == ACT_SILU_MUL and addend is None:
            raise RuntimeError("FusedLinear: ACT_SILU_MUL needs the multiplier in `addend`")
        add = None if (act != ACT_SILU_MUL and _is_zero_addend(addend)) else np.ascontiguousarray(_np(addend).reshape(M, -1)[:, :N])
        y = O.linear_fused(qx, qw, sx, sw, xo=xo, wo=wo, addend=add, bias=b, act=act, bit=bit)
    if row_amax is not None:                                 # the next layer's pre-pass maxima: |fp16 bits| over the columns its map does not mark
        if bit != 8 or row_amax.numel() < M or row_amax.element_size() != 4:
            raise RuntimeError("FusedLinear: row_amax needs an int8 GEMM and a 4-byte buffer of at least M entries")
        mag = (y.view(np.uint16) & 0x7fff).astype(np.int64)
        if col_mask is not None:
            cols = y.shape[1]
            m = _np(col_mask).view(np.uint32)
            marked = np.unpackbits(m[: (cols + 31) // 32].view(np.uint8), bitorder="little")[:cols].astype(bool)
            mag[:, marked] = 0
        cur = _np(row_amax).astype(np.int64)
        cur[:M] = np.maximum(cur[:M], mag.max(axis=1) if mag.shape[1] else 0)
        row_amax.copy_(torch.from_numpy(cur.astype(np.int32)))
    yt = torch.from_numpy(y)
    if out is not None:
        out.copy_(yt)
        return out
    return yt


class ForwardPlan:
    """The kept argument block of mixq_linear_forward, as the product's ForwardPlan (mixq_amd/mixlib.py) presents it."""

    def __init__(self, M, N, K, bit, sigma, ldx, ind_buf, n, n_dev, x_scale, q_w, scale_col, w_out, bias, qfmt, act=ACT_NONE, kept_mask=None):
        calls.append("ForwardPlan")
        if x_scale.numel() < M or scale_col.numel() < N:
            raise RuntimeError("ForwardPlan: x_scale / scale_col are shorter than M / N")
        self.n_cap = 0 if ind_buf is None else int(ind_buf.numel())
        if kept_mask is not None and (kept_mask.element_size() != 4 or kept_mask.numel() < kept_map_words(K)):
            raise RuntimeError("ForwardPlan: kept_mask must be the kept outlier map of K columns")
        self.kept_mask = kept_mask if self.n_cap else None
        self.keep = (ind_buf, n_dev, x_scale, q_w, scale_col, w_out, bias, kept_mask)
        self.M, self.N, self.K, self.bit, self.sigma, self.ldx, self.n, self.qfmt, self.act = M, N, K, bit, sigma, ldx, n, qfmt, act
        self.device = x_scale.device
        self.lock = threading.Lock()
        self.captured = False

    def run(self, x, row_amax=None, col_mask=None):
        calls.append("ForwardPlan.run")
        if x.dtype != torch.float16 or x.dim() != 2 or x.shape[0] != self.M or x.shape[1] != self.K or x.stride(1) != 1 or x.stride(0) != self.ldx:
            raise RuntimeError(f"ForwardPlan: x must be float16 [{self.M}, {self.K}] with row stride {self.ldx} and a contiguous last dimension")
        ind_buf, n_dev, x_scale, q_w, scale_col, w_out, bias, _ = self.keep
        cm = col_mask if col_mask is not None else self.kept_mask
        if row_amax is not None:
            # the rows' maxima were left by the producing GEMM: what the known-maximum quantiser reads (and clears) must be what the
            # two-pass quantiser would find over the columns this layer does not extract
            live, _ = _live(ind_buf, n_dev)
            xa = np.abs(_np(x).astype(np.float32))
            xa[:, live] = 0
            want = xa.max(axis=1).astype(np.float16).view(np.uint16).astype(np.int32) if xa.shape[1] else np.zeros(self.M, np.int32)
            assert np.array_equal(_np(row_amax)[: self.M], want), "row maxima handed over by the producer are not this input's"
            row_amax[: self.M] = 0
        q, xo = QuantFused(x, ind_buf, x_scale, self.bit, self.sigma, n_dev=n_dev, fmt=self.qfmt, col_mask=cm)
        calls.pop()
        y = FusedLinear(q, q_w, x_scale, scale_col, xo, None if w_out is None else w_out[:, : self.n_cap], self.n_cap, bias, self.M, self.N, self.K,
                        bit=self.bit, act=self.act, n_out_dev=n_dev)
        calls.pop()
        return y, q, (xo[:, : self.n] if xo is not None else None)


def layernorm_forward_cuda(x, weight, out, eps):
    calls.append("layernorm_forward_cuda")
    K = x.shape[-1]
    y = O.rmsnorm(_np(x.reshape(-1, K)), _np(weight), eps)
    out.reshape(-1, K).copy_(torch.from_numpy(y))
    return out


def RMSNormQuantFused(x, weight, out, eps, ind, x_scale, bit, sigma=6.0, flag=None, packed=False, fmt=None, n_dev=None, col_mask=None):
    calls.append("RMSNormQuantFused")
    fmt = _want_fmt(packed, fmt)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    M = x2.shape[0]
    if x_scale.numel() < M:
        raise RuntimeError(f"RMSNormQuantFused: x_scale holds {x_scale.numel()} rows, the batch has {M}")
    live, cap = _live(ind, n_dev)
    _check_kept_map(col_mask, live, K, "RMSNormQuantFused")
    y, xo, q, s = O.rmsnorm_quant(_np(x2), _np(weight), eps, live, bit)
    out.reshape(-1, K).copy_(torch.from_numpy(y))
    x_scale.reshape(-1)[0:M] = torch.from_numpy(s)
    if flag is not None and O.mispredicted(s, sigma, bit):
        flag |= 1
    x_out = None
    if cap:
        x_out = torch.zeros((M, (cap + 15) // 16 * 16), dtype=torch.float16)
        x_out[:, :live.size] = torch.from_numpy(xo)
    return _as_fmt(q, fmt, torch.int8 if bit == 8 else torch.uint8), (x_out[:, :cap] if cap else None)


def PackW8A16(q_weight_kn):
    calls.append("PackW8A16")
    return q_weight_kn                        # the oracle reads the checkpoint layout directly


def W8A16Linear(x, w_packed, scale_col, bias, N, K, out=None):
    calls.append("W8A16Linear")
    y = O.w8a16_linear(_np(x.reshape(-1, K)), _np(w_packed), _np(scale_col), None if bias is None else _np(bias))
    return torch.from_numpy(y)
