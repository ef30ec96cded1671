"""`mixlib`-compatible operator module backed by the gfx950 HIP kernels.

The reference's hot path calls an un-vendored CUDA extension named `mixlib` (call sites
/root/reference/mixquant/modules/linear.py:22,189-193,205,221,235-283,321-366).  This module exposes the SAME
function names with the SAME positional argument orders, so reference-style operator code runs unchanged on an
MI355X when `sys.modules['mixlib'] = mixq_amd.mixlib`.  Every function allocates its result
with torch (device memory / stream plumbing only) and forwards raw device pointers to the C ABI of
libmixq_hip.so (include/mixq_hip.h).  There is no CPU path: CPU tensors raise.

On top of the reference surface it offers the fused entry points the MI355X design adds (QuantFused,
DetectOutlierCols, DequantWeightCols, FusedLinear), used by mixq_amd.linear.MixLinear_GEMM.
"""
from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass, replace

import torch

from . import _capi
from ._capi import (ACT_NONE, ACT_SILU, ACT_SILU_MUL, ACT_SILU_PAIR, FMT_PLAIN, FMT_P16X64, FMT_F16X64, FMT_F6X128, FMT_R6X128, X_PACKED, W_PACKED, W_F16X64,
                    XW_F6X128)


def dev_check(*ts):
    """Every tensor handed to a kernel lives on the GPU: the product has no CPU path."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mixq_amd.mixlib: expected a GPU (HIP) tensor; there is no CPU fallback")


_dev_check = dev_check


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rows(t, what):
    """(ptr, row stride in elements) of a 2-D tensor whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"mixq_amd.mixlib: {what} must be 2-D with a contiguous last dimension")
    return t.data_ptr(), t.stride(0)


def _is_zero_addend(addend):
    """The reference passes MixLibCache.zeros (all zero, row stride 36864 != N) when there are no outliers.  Recognised by
    identity only (the `_mixq_all_zero` tag MixLibCache puts on it): a caller's own stride-0 broadcast addend is real data
    and goes to the kernel with lda = 0."""
    return addend is None or getattr(addend, "_mixq_all_zero", False)


def _check_ind(ind, what):
    if ind.dtype != torch.int32:
        raise RuntimeError(f"{what}: `ind` must be int32 (got {ind.dtype})")
    return ind if ind.is_contiguous() else ind.contiguous()


# ---- storage format of a quantised operand travels WITH the tensor -------------------------------------------------
# (a separate flag on the shared MixLibCache would describe whatever layer wrote it last, not the tensor handed over)
def set_fmt(t, fmt):
    t._mixq_fmt = fmt
    return t


def fmt_of(t):
    return getattr(t, "_mixq_fmt", FMT_PLAIN)


def _layout_bits(x_fmt, w_fmt):
    if x_fmt in (FMT_F6X128, FMT_R6X128) or w_fmt in (FMT_F6X128, FMT_R6X128):
        if x_fmt != FMT_R6X128 or w_fmt != FMT_F6X128:
            raise RuntimeError("mixq_amd.mixlib: the FP6 form of the W4A4 GEMM takes activations in R6X128 and weights in F6X128")
        return XW_F6X128
    if x_fmt == FMT_F16X64:
        raise RuntimeError("mixq_amd.mixlib: F16X64 is a weight format; the GEMMs take activations plain or in P16X64")
    return ({FMT_PLAIN: 0, FMT_P16X64: X_PACKED}[x_fmt] | {FMT_PLAIN: 0, FMT_P16X64: W_PACKED, FMT_F16X64: W_F16X64}[w_fmt])


# ------------------------------------------------------------------------------------------------------------
# reference surface
# ------------------------------------------------------------------------------------------------------------
# ---- opt-in: packed operands behind the UNCHANGED reference operator -------------------------------------------------
# The reference's forward treats `cache.q_xcache` and `self.q_weight` as opaque handles between mixlib calls (linear.py:190-283).
# With configure(packed_operands=True), FindRowScale hands back an [M, KB] view of a buffer that really holds the P16X64 image (the
# tag rides on the returned tensor object) and the GEMM entry points use a fragment-order copy of every plain weight they are
# given (cached per tensor: +1x the weight bytes) - the kernels of the native operator, behind the reference's own code.
# Off by default: a caller that slices, copies or inspects q_x would lose the tag or read tile-major bytes.
_wpacked = {}          # (data_ptr, version, shape) -> (weight, fragment-order copy); the entry pins the weight: see eetq._packed


# The three tensor-subclass handles below (and PendingGemmI32) sit on torch's __torch_function__ protocol and two private helpers
# (torch.Tensor._make_subclass, torch._C.DisableTorchFunctionSubclass).  They were written and tested against this torch series; on
# another one they are still opt-in, and say so once.
_TESTED_TORCH = "2.10"
_warned_torch = False


def _check_torch_for_handles():
    global _warned_torch
    if not _warned_torch and not torch.__version__.startswith(_TESTED_TORCH):
        import warnings
        _warned_torch = True
        warnings.warn(f"mixq_amd.mixlib: the deferred-tensor switches (mixlib.configure: lazy_gemm / packed_operands / fused_outliers / "
                      f"fused_prepass) were tested with torch {_TESTED_TORCH}.x; this is torch {torch.__version__}", RuntimeWarning, stacklevel=3)


@dataclass(frozen=True)
class ShimConfig:
    """How this module serves the reference's UNCHANGED call sequence - ONE object, handed over once by the
    maintainer who binds the module (`mixlib.configure(...)` next to `sys.modules["mixlib"] = mixq_amd.mixlib`), instead of round 5's
    four independent module switches.  Every field keeps every result the reference's own Python observes; the caveat of each is why it
    is a choice:

      lazy_gemm        `gemm` returns a deferred int32 product (PendingGemmI32) so that the reference's `gemm` + `dequantizeInt8[Silu]` pair
                       (its arch == 9 route, which gfx950 takes: linear.py:232-241) runs as ONE fused launch; any other use of the handle
                       computes the exact int32 product first.  ON by default: the only consumer that could see uncomputed storage is a
                       foreign C++ extension taking the at::Tensor, and the reference has none on this path.
      packed_operands  FindRowScale hands back q_x as an [M, KB] view of a tile-major (P16x64) buffer, weights are re-tiled once into
                       fragment order (+1x the weight bytes, cached per storage).  Off by default: a caller that slices or copies q_x
                       loses the tag.
      fused_outliers   the `torch.mm(activation_outliers, weight_cache.T)` of linear.py:236,248 becomes the fp16 MFMA tail of the GEMM.
                       Off by default because ONE rounding changes (the literal route rounds the product to fp16 first: <= 2 fp16 ulp).
      fused_prepass    (needs fused_outliers) ExtractOutliersAndSetToZeros + FindRowScale on the same tensor run as the one-pass kernel.
                       Off by default: code that READS x between the two calls would still see the outlier columns unzeroed.
    All three opt-ins on = the native operator's speed behind the reference's own code (profiles/r05_arch9_route.txt)."""
    lazy_gemm: bool = True
    packed_operands: bool = False
    fused_outliers: bool = False
    fused_prepass: bool = False

    @classmethod
    def fastest(cls):
        return cls(lazy_gemm=True, packed_operands=True, fused_outliers=True, fused_prepass=True)

    @classmethod
    def literal(cls):
        """Every reference call = the kernel of that name, nothing deferred (the debugging route)."""
        return cls(lazy_gemm=False)


_cfg = ShimConfig()


def configure(config=None, **fields):
    """Install the shim's configuration (a ShimConfig, and / or single fields by name) and return the previous one:
        mixlib.configure(mixlib.ShimConfig.fastest())        # once, where the module is bound
        prev = mixlib.configure(fused_outliers=True); ...; mixlib.configure(prev)
    A deferred extraction still pending under the old configuration runs first; caches that only serve a switched-off route are dropped."""
    global _cfg
    new = replace(config if config is not None else _cfg, **fields)
    if new.fused_prepass and not new.fused_outliers:
        raise ValueError("mixlib.configure: fused_prepass needs fused_outliers (the outlier tensor must be one of the shim's)")
    if new.packed_operands or new.fused_outliers or new.fused_prepass or (new.lazy_gemm and not ShimConfig().lazy_gemm):
        _check_torch_for_handles()
    _flush_pending_extract()
    prev, _cfg = _cfg, new
    if not new.packed_operands:
        _wpacked.clear()
    if not new.fused_outliers:
        _wo_padded.clear()
    return prev


@contextmanager
def configured(config=None, **fields):
    """`with mixlib.configured(fused_outliers=True): ...` - a scoped configure (tests, tools)."""
    prev = configure(config, **fields)
    try:
        yield _cfg
    finally:
        configure(prev)


def _weight_for_gemm(q_w, bit):
    if not _cfg.packed_operands or fmt_of(q_w) != FMT_PLAIN or q_w.dim() != 2 or q_w.shape[1] % 64 or not q_w.is_contiguous():
        return q_w
    key = (q_w.data_ptr(), q_w._version, tuple(q_w.shape))
    e = _wpacked.get(key)
    if e is None:
        if len(_wpacked) > 1024:
            _wpacked.clear()
        e = _wpacked[key] = (q_w, PackOperand(q_w, FMT_F16X64 if bit == 8 else FMT_P16X64))
    return e[1]


def FindRowScale(x, x_scale, M, K, bit=8):
    """mixlib.FindRowScale(x, x_scale, M, K, bit) -> q_x  (linear.py:190-193).  Writes x_scale[0:M] in place."""
    _dev_check(x, x_scale)
    if x.dtype != torch.float16 or x_scale.dtype != torch.float16:
        raise RuntimeError("FindRowScale: x and x_scale must be float16")
    x2 = x.reshape(-1, x.shape[-1])
    xp, ldx = _rows(x2, "x")
    if x2.shape[0] < M or x2.shape[1] != K or x_scale.numel() < M:
        raise RuntimeError("FindRowScale: shape mismatch")
    if bit not in (4, 8):
        raise RuntimeError("FindRowScale: bit must be 4 or 8")
    KB, dt = (K, torch.int8) if bit == 8 else (K // 2, torch.uint8)
    global _pending_extract
    pend = _pending_extract
    if pend is not None:
        ind_p, x_p, ldo_p = pend._mixq_pending
        same = (x_p.data_ptr() == x2.data_ptr() and tuple(x_p.shape) == tuple(x2.shape) and x_p.stride(0) == x2.stride(0)
                and x2.shape[0] == M and K % 8 == 0)
        if same:                                             # extract + zero + scale + quantise in ONE pass (the native operator's kernel)
            _pending_extract = None
            pend.__dict__.pop("_mixq_pending", None)
            packed = _cfg.packed_operands and M > 0 and KB % 64 == 0
            q = torch.empty((packed_rows(M) if packed else M, KB), dtype=dt, device=x.device)
            with torch._C.DisableTorchFunctionSubclass():
                _capi.call("mixq_quant_fused", xp, ind_p.data_ptr(), ind_p.numel(), None, x_scale.data_ptr(), q.data_ptr(), pend.data_ptr(),
                           None, M, K, ldx, ldo_p, bit, 6.0, FMT_P16X64 if packed else FMT_PLAIN, _stream())
            return set_fmt(q[:M], FMT_P16X64) if packed else q
        _flush_pending_extract()
    if _cfg.packed_operands and M > 0 and KB % 64 == 0:
        buf = torch.empty((packed_rows(M), KB), dtype=dt, device=x.device)
        _capi.call("mixq_find_row_scale", xp, x_scale.data_ptr(), buf.data_ptr(), M, K, ldx, bit, FMT_P16X64, _stream())
        return set_fmt(buf[:M], FMT_P16X64)              # same storage, same base pointer: an opaque handle of the reference's shape
    q = torch.empty((M, KB), dtype=dt, device=x.device)
    _capi.call("mixq_find_row_scale", xp, x_scale.data_ptr(), q.data_ptr(), M, K, ldx, bit, FMT_PLAIN, _stream())
    return q


# ---- opt-in: the outlier product fused into the GEMM behind the UNCHANGED reference operator ------------------------------
# The reference computes `outliers_fp16 = torch.mm(cache.activation_outliers, self.weight_cache.T)` (linear.py:236,248) - an [M,N] fp16
# tensor written by one kernel and read back as the addend of the next - where the native operator runs the same product as the fp16
# MFMA tail of the int8 GEMM.  With configure(fused_outliers=True) ExtractOutliersAndSetToZeros returns its (real) [M,n] tensor as an
# OutlierActivations; torch.mm / matmul / @ of THAT tensor with a transposed fp16 [N,n] matrix yields a PendingOutlierProduct: an [M,N]
# fp16 tensor whose values are only computed if something other than a mixlib GEMM entry point looks at them.  The GEMM entry points
# take the two factors as their outlier operands instead.  Off by default because it changes one rounding: the literal route rounds
# the product to fp16 before it is added (torch.mm), the fused tail adds it in fp32 like the native operator (include/mixq_hip.h,
# convention 2; <= 2 fp16 ulp apart).
_wo_padded = {}        # (data_ptr, version, shape, stride) -> (weight_cache view, copy with a 16-column-padded pitch); pins its source


def _pad16(n):
    return (n + 15) // 16 * 16


# ---- opt-in: extract + quantise as ONE pass behind the reference's two calls ----------------------------------------------
# The reference calls ExtractOutliersAndSetToZeros(ind, x) and then FindRowScale(x, ...) on the same tensor (linear.py:189-193):
# two kernels, two passes over X, two launch floors.  With configure(fused_prepass=True) (needs configure(fused_outliers=True): the outlier
# tensor must be one of ours) the first call only allocates its result and remembers (ind, x); FindRowScale on that same x then runs
# the native operator's fused kernel, which gathers, zeroes, scales and quantises in one pass.  Anything else that touches the outlier
# tensor first - or a FindRowScale / ExtractOutliersAndSetToZeros on another tensor - runs the deferred extraction at once, in order.
# What it cannot see: code that READS x between the two calls would still find the outlier columns unzeroed.  Off by default.
_pending_extract = None        # the OutlierActivations whose extraction is still deferred (a STRONG reference: the in-place zeroing of
                               # x is a side effect the caller is owed even if it drops the returned tensor before FindRowScale runs)


def _flush_pending_extract():
    global _pending_extract
    t = _pending_extract
    _pending_extract = None
    if t is not None:
        t._run_extract()


class OutlierActivations(torch.Tensor):
    """x_out [M,n] as returned by ExtractOutliersAndSetToZeros under configure(fused_outliers=True): a view of 16-column-padded storage
    (what the GEMM's tail reads) that recognises the reference's torch.mm with weight_cache.T."""

    def _run_extract(self):
        """The deferred extraction of ShimConfig.fused_prepass (a no-op when there is none)."""
        pend = self.__dict__.pop("_mixq_pending", None)
        if pend is not None:
            ind, x, ldo = pend
            with torch._C.DisableTorchFunctionSubclass():
                _capi.call("mixq_extract_outliers_zero", x.data_ptr(), ind.data_ptr(), ind.numel(), self.data_ptr(), x.shape[0], x.shape[1],
                           x.stride(0), ldo, _stream())

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        T = torch.Tensor
        if func not in _meta_funcs():
            for a in list(args) + list(kwargs.values()):
                for b in (a if isinstance(a, (list, tuple)) else (a,)):
                    if isinstance(b, OutlierActivations) and "_mixq_pending" in b.__dict__:
                        _flush_pending_extract()
                        b._run_extract()
        if _cfg.fused_outliers and not kwargs and len(args) == 2 and func in (torch.mm, torch.matmul, T.mm, T.matmul, T.__matmul__):
            a, b = args
            if (isinstance(a, OutlierActivations) and isinstance(b, torch.Tensor) and not isinstance(b, OutlierActivations)
                    and a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[0] and a.shape[1] > 0
                    and a.dtype == torch.float16 and b.dtype == torch.float16 and b.device == a.device and b.stride(0) == 1):
                return PendingOutlierProduct(a, b)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class PendingOutlierProduct(torch.Tensor):
    """torch.mm(activation_outliers, weight_cache.T), deferred: see ShimConfig.fused_outliers."""

    @staticmethod
    def __new__(cls, xo, wo_t):
        t = torch.Tensor._make_subclass(cls, torch.empty((xo.shape[0], wo_t.shape[1]), dtype=torch.float16, device=xo.device))
        t._mixq_args = (xo, wo_t)
        t._mixq_done = False
        return t

    def _materialize(self):
        if getattr(self, "_mixq_done", True):
            return
        self._mixq_done = True
        with torch._C.DisableTorchFunctionSubclass():
            xo, wo_t = self._mixq_args
            torch.mm(xo.as_subclass(torch.Tensor), wo_t, out=self.as_subclass(torch.Tensor))
        self._mixq_args = None

    # exports that do not go through __torch_function__ compute the values first
    def __dlpack__(self, *args, **kwargs):
        self._materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor).__dlpack__(*args, **kwargs)

    @property
    def __cuda_array_interface__(self):
        self._materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor).__cuda_array_interface__

    def _operands(self):
        """(x_out, ldxo-ready tensor, w_out padded, n) for the fused tail, or None once materialised."""
        if getattr(self, "_mixq_done", True):
            return None
        xo, wo_t = self._mixq_args
        if isinstance(xo, OutlierActivations) and "_mixq_pending" in xo.__dict__:
            _flush_pending_extract()
            xo._run_extract()
        n = xo.shape[1]
        wo = wo_t.t()                                        # [N, n], unit column stride
        pad = _pad16(n)
        with torch._C.DisableTorchFunctionSubclass():
            xo = xo.as_subclass(torch.Tensor)
            if xo.stride(1) != 1 or xo.stride(0) % 8 or xo.stride(0) < pad or xo.data_ptr() % 16:
                return None                                  # not the padded storage ExtractOutliersAndSetToZeros hands out
            if wo.stride(0) % 8 or wo.stride(0) < pad or wo.data_ptr() % 16:
                key = (wo.data_ptr(), wo._version, tuple(wo.shape), wo.stride(0))
                e = _wo_padded.get(key)
                wp = e[1] if e is not None else None
                if wp is None:
                    if len(_wo_padded) > 1024:
                        _wo_padded.clear()
                    buf = torch.zeros((wo.shape[0], pad), dtype=torch.float16, device=wo.device)
                    buf[:, :n] = wo
                    _wo_padded[key] = (wo, buf)
                    wp = buf
                wo = wp[:, :n]
        return xo, wo, n

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func not in _meta_funcs():
            for a in list(args) + list(kwargs.values()):
                if isinstance(a, PendingOutlierProduct):
                    a._materialize()
                elif isinstance(a, (list, tuple)):
                    for b in a:
                        if isinstance(b, PendingOutlierProduct):
                            b._materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def ExtractOutliersAndSetToZeros(ind, x):
    """mixlib.ExtractOutliersAndSetToZeros(ind, x) -> x_out [M,n]; zeroes columns `ind` of x IN PLACE (linear.py:189,205)."""
    global _pending_extract
    _dev_check(ind, x)
    if x.dtype != torch.float16:
        raise RuntimeError("ExtractOutliersAndSetToZeros: x must be float16")
    ind = _check_ind(ind, "ExtractOutliersAndSetToZeros")
    xp, ldx = _rows(x, "x")
    M, K = x.shape
    n = ind.numel()
    _flush_pending_extract()                                 # (an earlier deferred extraction runs first: program order)
    if _cfg.fused_outliers and n:
        buf = torch.empty((M, _pad16(n)), dtype=torch.float16, device=x.device)          # (columns >= n: never read as values)
        out = buf[:, :n].as_subclass(OutlierActivations)
        if _cfg.fused_prepass and x.dim() == 2:
            out._mixq_pending = (ind, x, buf.shape[1])
            _pending_extract = out
            return out
        _capi.call("mixq_extract_outliers_zero", xp, ind.data_ptr(), n, buf.data_ptr(), M, K, ldx, buf.shape[1], _stream())
        return out
    out = torch.empty((M, n), dtype=torch.float16, device=x.device)
    if n:
        _capi.call("mixq_extract_outliers_zero", xp, ind.data_ptr(), n, out.data_ptr(), M, K, ldx, n, _stream())
    return out


def _outlier_operands(addend):
    """(x_out ptr, ldxo, w_out ptr, ldwo, n, keepalive) when `addend` is a still-deferred outlier product, else None."""
    if isinstance(addend, PendingOutlierProduct):
        ops = addend._operands()
        if ops is not None:
            xo, wo, n = ops
            return xo.data_ptr(), xo.stride(0), wo.data_ptr(), wo.stride(0), n, (xo, wo)
    return None


def _fused_dequant(q_x, q_w, x_scale, scale_col, addend, M, N, K, bit, act):
    _dev_check(q_x, q_w, x_scale, scale_col)
    if fmt_of(q_x) != FMT_PLAIN:
        q_w = _weight_for_gemm(q_w, bit)                 # (packed activations need packed weights: ShimConfig.packed_operands)
        if fmt_of(q_w) == FMT_PLAIN:
            raise RuntimeError("mixq_amd.mixlib: packed q_x with a plain weight that cannot be packed (K % 64)")
    y = torch.empty((M, N), dtype=torch.float16, device=q_x.device)
    xop, ldxo, wop, ldwo, n_out = None, 0, None, 0, 0
    outl = _outlier_operands(addend)
    if outl is not None and tuple(addend.shape) == (M, N):
        xop, ldxo, wop, ldwo, n_out, _keep = outl            # the product runs as the GEMM's fp16 tail; nothing is added afterwards
        ap, lda = None, 0
    elif _is_zero_addend(addend):
        ap, lda = None, 0
    else:
        if isinstance(addend, PendingOutlierProduct):
            addend._materialize()
        _dev_check(addend)
        ap, lda = _rows(addend, "addend")
    fn = "mixq_gemm_i8_fused" if bit == 8 else "mixq_gemm_i4_fused"
    _capi.call(fn, q_x.data_ptr(), q_w.data_ptr(), x_scale.data_ptr(), scale_col.data_ptr(),
               xop, ldxo, wop, ldwo, n_out, None, ap, lda, None, y.data_ptr(), N, M, N, K, act, _layout_bits(fmt_of(q_x), fmt_of(q_w)),
               _stream())
    return y


def int8FusedDequantize(q_x, q_w, x_scale, scale_col, addend, M, N, K):
    """mixlib.int8FusedDequantize (linear.py:251-256, 268-273)."""
    return _fused_dequant(q_x, q_w, x_scale, scale_col, addend, M, N, K, 8, ACT_NONE)


def int8FusedDequantizeSilu(q_x, q_w, x_scale, scale_col, addend, M, N, K):
    """mixlib.int8FusedDequantizeSilu (linear.py:337-351)."""
    return _fused_dequant(q_x, q_w, x_scale, scale_col, addend, M, N, K, 8, ACT_SILU)


def int4FusedDequantize(q_x, q_w, x_scale, scale_col, addend, M, N, K_half):
    """mixlib.int4FusedDequantize: the last argument is K/2 (linear.py:259-265)."""
    return _fused_dequant(q_x, q_w, x_scale, scale_col, addend, M, N, 2 * K_half, 4, ACT_NONE)


def int4FusedDequantizeSilu(q_x, q_w, x_scale, scale_col, addend, M, N, K_half):
    """mixlib.int4FusedDequantizeSilu (linear.py:360-366)."""
    return _fused_dequant(q_x, q_w, x_scale, scale_col, addend, M, N, 2 * K_half, 4, ACT_SILU)


_META_FUNCS = None


def _meta_funcs():
    """torch functions that only read metadata of a tensor: they do not force a pending product to be computed."""
    global _META_FUNCS
    if _META_FUNCS is None:
        T = torch.Tensor
        fs = {T.size, T.dim, T.numel, T.stride, T.element_size, T.is_contiguous, T.nelement, T.ndimension, T.storage_offset,
              T.is_floating_point, T.is_complex, T.get_device, T.__len__}
        for prop in ("shape", "dtype", "device", "ndim", "is_cuda", "layout", "requires_grad", "grad_fn", "is_leaf", "is_sparse",
                     "is_quantized", "is_meta", "names"):
            fs.add(getattr(T, prop).__get__)
        _META_FUNCS = fs
    return _META_FUNCS


class PendingGemmI32(torch.Tensor):
    """The int32 [M,N] tensor `mixlib.gemm` returns, with the product itself deferred.

    On gfx950 torch reports capability major 9, so the UNCHANGED reference takes its `arch == 9` branch
    (linear.py:86,234-241,320-327): `y = mixlib.gemm(...)` followed by `mixlib.dequantizeInt8[Silu](y, x_scale, scale_col,
    addend, 8, M, N)`.  Run literally, that is a 4*M*N-byte int32 round trip through HBM plus a second kernel.  Here `gemm`
    only allocates the result and remembers its operands; `dequantizeInt8[Silu]` recognises the handle and launches the ONE
    fused kernel (same arithmetic: exact int32 accumulator, fp32 epilogue, one rounding).  Any other use of the tensor - any
    torch function that is not a pure metadata query - first computes the int32 product into its storage, so code that
    really wants the integers gets them."""

    @staticmethod
    def __new__(cls, q_x, q_w, M, N, K):
        t = torch.Tensor._make_subclass(cls, torch.empty((M, N), dtype=torch.int32, device=q_x.device))
        t._mixq_args = (q_x, q_w, M, N, K)
        t._mixq_done = False
        return t

    def _materialize(self):
        if getattr(self, "_mixq_done", True):
            return
        self._mixq_done = True
        with torch._C.DisableTorchFunctionSubclass():
            q_x, q_w, M, N, K = self._mixq_args
            if fmt_of(q_x) != FMT_PLAIN:                     # ShimConfig.packed_operands: the raw kernel wants the plain matrix back
                q_x = UnpackOperand(q_x.as_strided((packed_rows(M), K), (K, 1), q_x.storage_offset()), M, fmt_of(q_x))
            _capi.call("mixq_gemm_i8", q_x.data_ptr(), q_w.data_ptr(), self.data_ptr(), N, M, N, K, _stream())
        self._mixq_args = None

    # exports that do not go through __torch_function__ compute the values first
    def __dlpack__(self, *args, **kwargs):
        self._materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor).__dlpack__(*args, **kwargs)

    @property
    def __cuda_array_interface__(self):
        self._materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor).__cuda_array_interface__

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func not in _meta_funcs():
            for a in list(args) + list(kwargs.values()):
                if isinstance(a, PendingGemmI32):
                    a._materialize()
                elif isinstance(a, (list, tuple)):
                    for b in a:
                        if isinstance(b, PendingGemmI32):
                            b._materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def _lazy():
    return _cfg.lazy_gemm or _cfg.packed_operands or _cfg.fused_outliers


def gemm(q_x, q_w, M, N, K):
    """mixlib.gemm(q_x, q_w, M, N, K) -> int32 [M,N] (linear.py:235,321)."""
    _dev_check(q_x, q_w)
    if fmt_of(q_w) != FMT_PLAIN or (fmt_of(q_x) != FMT_PLAIN and not _lazy()):
        raise RuntimeError("mixlib.gemm: the raw int32 GEMM takes plain row-major operands")
    if K % 64 or N % 4:
        raise _capi.MixqError("mixq_gemm_i8", _capi.MIXQ_ESHAPE)
    if q_x.dim() != 2 or q_w.dim() != 2 or q_x.shape[0] < M or q_x.shape[1] != K or q_w.shape[0] < N or q_w.shape[1] != K \
            or q_x.element_size() != 1 or q_w.element_size() != 1:
        # (the reference's arch == 9 route calls this for 4-bit layers too, linear.py:235, with nibble-packed [.,K/2] operands:
        # a one-byte-per-value GEMM over them would read past both buffers)
        raise RuntimeError(f"mixlib.gemm: operands must be one-byte [>=M,K] and [>=N,K] matrices (got {tuple(q_x.shape)}, "
                           f"{tuple(q_w.shape)} for M,N,K = {M},{N},{K}); nibble-packed int4 operands go through int4FusedDequantize")
    if _lazy() and M > 0 and N > 0:
        return PendingGemmI32(q_x, q_w, M, N, K)
    y = torch.empty((M, N), dtype=torch.int32, device=q_x.device)
    _capi.call("mixq_gemm_i8", q_x.data_ptr(), q_w.data_ptr(), y.data_ptr(), N, M, N, K, _stream())
    return y


def _dequant(y32, x_scale, scale_col, addend, bit, M, N, act):
    _dev_check(y32, x_scale, scale_col)
    if isinstance(y32, PendingGemmI32) and not y32._mixq_done and tuple(y32._mixq_args[2:4]) == (M, N):
        q_x, q_w, _, _, K = y32._mixq_args                  # the product was never needed on its own: one fused kernel
        return _fused_dequant(q_x, q_w, x_scale, scale_col, addend, M, N, K, 8, act)
    y = torch.empty((M, N), dtype=torch.float16, device=y32.device)
    if _is_zero_addend(addend):
        ap, lda = None, 0
    else:
        if isinstance(addend, PendingOutlierProduct):
            addend._materialize()                            # the literal pair adds the product as the reference computed it
        ap, lda = _rows(addend, "addend")
    yp, ldy32 = _rows(y32, "y32")
    _capi.call("mixq_dequant", yp, ldy32, x_scale.data_ptr(), scale_col.data_ptr(), ap, lda, None, y.data_ptr(), N, M, N,
               act, _stream())
    return y


def dequantizeInt8(y32, x_scale, scale_col, addend, bit, M, N):
    """mixlib.dequantizeInt8 (linear.py:238,241)."""
    return _dequant(y32, x_scale, scale_col, addend, bit, M, N, ACT_NONE)


def dequantizeInt8Silu(y32, x_scale, scale_col, addend, bit, M, N):
    """mixlib.dequantizeInt8Silu (linear.py:324,327)."""
    return _dequant(y32, x_scale, scale_col, addend, bit, M, N, ACT_SILU)


def unpack_int4_to_fp16(q_w, ind):
    """mixlib.unpack_int4_to_fp16(q_w, ind) -> fp16 [N,n] sign-extended int4 weight columns (linear.py:20-22)."""
    _dev_check(q_w, ind)
    ind = _check_ind(ind, "unpack_int4_to_fp16")
    if not q_w.is_contiguous():
        raise RuntimeError("unpack_int4_to_fp16: q_w must be contiguous")
    N, Kh = q_w.shape
    n = ind.numel()
    out = torch.empty((N, n), dtype=torch.float16, device=q_w.device)
    if n:
        _capi.call("mixq_dequant_weight_cols", q_w.data_ptr(), None, ind.data_ptr(), n, out.data_ptr(), N, 2 * Kh, n, 4, _stream())
    return out


# ------------------------------------------------------------------------------------------------------------
# MI355X additions (fused forms)
# ------------------------------------------------------------------------------------------------------------
def packed_rows(rows):
    return (rows + 15) // 16 * 16


def PackOperand(q, fmt=FMT_P16X64):
    """Re-tile a plain [R,KB] int8/uint8 operand into a packed layout (include/mixq_hip.h).  Returns a [roundup(R,16), KB]
    tensor of the same dtype holding the packed bytes (NOT addressable as a matrix), tagged with its format."""
    _dev_check(q)
    if q.dim() != 2 or not q.is_contiguous() or q.element_size() != 1:
        raise RuntimeError("PackOperand: expected a contiguous 2-D int8/uint8 tensor")
    R, KB = q.shape
    # (F6X128: KB = K / 2 bytes of nibbles per row in, 3 K / 4 bytes of FP6 codes per row out)
    out = torch.empty((packed_rows(R), KB * 3 // 2 if fmt in (FMT_F6X128, FMT_R6X128) else KB), dtype=q.dtype, device=q.device)
    _capi.call("mixq_pack_operand", q.data_ptr(), out.data_ptr(), R, KB, fmt, _stream())
    return set_fmt(out, fmt)


def EnsureWorkspace(device):
    """Register (once per process and device) the zero-filled workspace the stream-K and pairwise split-K forms of the GEMM hand partial tiles
    over through (mixq_gemm_set_workspace); without it the library runs its other tilings."""
    return _capi.ensure_workspace(device)


def PackP16x64(q):
    return PackOperand(q, FMT_P16X64)


def UnpackOperand(packed, R, fmt=None):
    """Inverse of PackOperand: the plain [R,KB] matrix of a packed image."""
    _dev_check(packed)
    fmt = fmt_of(packed) if fmt is None else fmt
    if fmt not in (FMT_P16X64, FMT_F16X64, FMT_F6X128, FMT_R6X128):
        raise RuntimeError("UnpackOperand: the tensor carries no packed-format tag; pass fmt")
    KB = packed.shape[1] * 2 // 3 if fmt in (FMT_F6X128, FMT_R6X128) else packed.shape[1]
    out = torch.empty((R, KB), dtype=packed.dtype, device=packed.device)
    _capi.call("mixq_unpack_operand", packed.data_ptr(), out.data_ptr(), R, KB, fmt, _stream())
    return out


def _q_row_bytes(K, bit, fmt):
    """Bytes per row of a quantised activation buffer: K int8, K / 2 nibbles, or 3 K / 4 FP6 codes."""
    if fmt in (FMT_F6X128, FMT_R6X128):
        if bit != 4:
            raise RuntimeError("mixq_amd.mixlib: F6X128 / R6X128 are int4 formats")
        return K * 3 // 4
    return K if bit == 8 else K // 2


def _want_fmt(packed, fmt):
    if fmt is not None:
        return fmt
    return FMT_P16X64 if packed else FMT_PLAIN


def kept_map_words(K):
    """int32 words of the kept outlier map of K columns (include/mixq_hip.h: bits, count, pad to 4 words, K 16-bit AND-masks)."""
    return (((K + 31) // 32 + 1 + 3) // 4) * 4 + ((K + 7) // 8) * 4


def _check_kept_map(col_mask, K, who):
    """A kept outlier map handed to a kernel: 4-byte words, contiguous, 16-byte aligned, and LONG ENOUGH for the layout the kernels read (round
    4's map stopped behind the count word; the kernels read the per-column AND-masks up to 2 K bytes further on - ADVICE r05)."""
    if col_mask.element_size() != 4 or col_mask.numel() < kept_map_words(K) or not col_mask.is_contiguous() or col_mask.data_ptr() % 16:
        raise RuntimeError(f"{who}: col_mask must be the 16-byte aligned kept outlier map of K = {K} columns - {kept_map_words(K)} int32 words "
                           f"(bit words, count word, pad, K 16-bit AND-masks; linear.kept_outlier_map), got {col_mask.numel()} x {col_mask.element_size()} bytes")


def QuantFused(x, ind, x_scale, bit, sigma, x_out=None, flag=None, n_dev=None, packed=False, fmt=None, col_mask=None):
    """(i)+(ii) in one pass over X: extract/zero the known outlier columns `ind`, per-row scale into x_scale[0:M],
    quantise, and raise the device-side misprediction flag.  Returns (q_x, x_out_view[M,n]).  With fmt = FMT_P16X64 /
    FMT_F16X64 (or packed=True: P16x64) q_x is emitted directly in that layout ([roundup(M,16), KB] bytes) and tagged.
    `n_dev` (device int32[1]) overrides the count: `ind` then is a buffer of capacity ind.numel().
    `col_mask`: the caller's kept outlier map of the live `ind` entries (device int32 words - bit words, count word, per-column AND-masks:
    include/mixq_hip.h, linear.kept_outlier_map) - the pass then needs ONE memory round trip (mixq_quant_fused_masked; same bytes out)."""
    _dev_check(x, x_scale, ind, col_mask)
    fmt = _want_fmt(packed, fmt)
    xp, ldx = _rows(x, "x")
    M, K = x.shape
    if x.dtype != torch.float16:
        raise RuntimeError("QuantFused: x must be float16")
    if x_scale.numel() < M:
        raise RuntimeError(f"QuantFused: x_scale holds {x_scale.numel()} rows, the batch has {M} (MixLibCache.inputdim too small)")
    n = 0 if ind is None else ind.numel()
    q = torch.empty((packed_rows(M) if fmt else M, _q_row_bytes(K, bit, fmt)),
                    dtype=torch.int8 if bit == 8 else torch.uint8, device=x.device)
    if n:
        ind = _check_ind(ind, "QuantFused")
        if x_out is None:
            ldo = (n + 15) // 16 * 16
            x_out = torch.empty((M, ldo), dtype=torch.float16, device=x.device)
        op, ldo = _rows(x_out, "x_out")
        ip = ind.data_ptr()
    else:
        op, ldo, ip = None, 0, None
    if col_mask is not None and n:
        _check_kept_map(col_mask, K, "QuantFused")
        _capi.call("mixq_quant_fused_masked", xp, ip, n, _ptr(n_dev), col_mask.data_ptr(), col_mask.numel(), x_scale.data_ptr(), q.data_ptr(), op, _ptr(flag), M, K,
                   ldx, ldo, bit, float(sigma), fmt, _stream())
    else:
        _capi.call("mixq_quant_fused", xp, ip, n, _ptr(n_dev), x_scale.data_ptr(), q.data_ptr(), op, _ptr(flag), M, K, ldx,
                   ldo, bit, float(sigma), fmt, _stream())
    return set_fmt(q, fmt), (x_out[:, :n] if n else None)


def FindRowScalePacked(x, x_scale, M, K, bit=8, fmt=FMT_P16X64):
    """FindRowScale emitting a packed layout."""
    _dev_check(x, x_scale)
    xp, ldx = _rows(x, "x")
    if x_scale.numel() < M:
        raise RuntimeError(f"FindRowScalePacked: x_scale holds {x_scale.numel()} rows, the batch has {M}")
    q = torch.empty((packed_rows(M), _q_row_bytes(K, bit, fmt)), dtype=torch.int8 if bit == 8 else torch.uint8, device=x.device)
    _capi.call("mixq_find_row_scale", xp, x_scale.data_ptr(), q.data_ptr(), M, K, ldx, bit, fmt, _stream())
    return set_fmt(q, fmt)


def DetectOutlierCols(x, sigma, scratch=None):
    """Device-side FindOutliers (linear.py:157-161): returns (ind_buf int32 [K], count int32 [1]) on device; the
    caller reads `count` (a host sync, warm-up only) and slices."""
    _dev_check(x)
    xp, ldx = _rows(x, "x")
    M, K = x.shape
    if scratch is None:
        scratch = (torch.empty(K, dtype=torch.uint8, device=x.device),
                   torch.empty(K, dtype=torch.int32, device=x.device),
                   torch.zeros(1, dtype=torch.int32, device=x.device))
    flags, ind_buf, count = scratch
    _capi.call("mixq_detect_outlier_cols", xp, float(sigma), flags.data_ptr(), ind_buf.data_ptr(), count.data_ptr(), M, K,
               ldx, _stream())
    return ind_buf, count


def DequantWeightCols(q_w, scale_col, ind, bit, out=None):
    """`q_weight[:,ind].to(fp16) * scale_col.T` (linear.py:207) / the int4 twin (linear.py:209-210) as one kernel."""
    _dev_check(q_w, scale_col, ind)
    ind = _check_ind(ind, "DequantWeightCols")
    if fmt_of(q_w) != FMT_PLAIN or q_w.dim() != 2 or not q_w.is_contiguous():
        raise RuntimeError("DequantWeightCols: q_w must be the plain, contiguous [N,KB] matrix")
    N = q_w.shape[0]
    K = q_w.shape[1] * (1 if bit == 8 else 2)
    n = ind.numel()
    if out is None:
        out = torch.empty((N, n), dtype=torch.float16, device=q_w.device)
    op, ldo = _rows(out, "out")
    if n:
        _capi.call("mixq_dequant_weight_cols", q_w.data_ptr(), scale_col.data_ptr(), ind.data_ptr(), n, op, N, K, ldo, bit, _stream())
    return out


PAIR_LAUNCH = True                       # FusedLinear serves ACT_SILU_PAIR (fused.MixLlamaMLP asks before taking its joint gate / up route)


def amax_supported(M, N, K, x_fmt, w_fmt):
    """True when the GEMM can leave the next layer's row maxima as a side output for this problem (mixq_gemm_amax_supported)."""
    if x_fmt != FMT_P16X64 or w_fmt != FMT_F16X64:
        return False
    return bool(_capi.load().mixq_gemm_amax_supported(M, N, K, X_PACKED | W_F16X64))


def FusedLinear(q_x, q_w, x_scale, scale_col, x_out, w_out, n_out, bias, M, N, K, bit=8, act=ACT_NONE, n_out_dev=None,
                addend=None, out=None, x_packed=None, w_packed=None, row_amax=None, col_mask=None):
    """(iii)+(iv): int8/int4 MFMA GEMM + dequant + fp16 outlier correction + addend + act + bias -> fp16 [M,N].
    The operand layouts come from the tensors' own format tags (set_fmt / the packing and quantising functions above);
    x_packed / w_packed = True/False force P16x64 / plain for untagged buffers.  `n_out_dev` (device int32[1]) overrides the
    outlier count: n_out then is the capacity of x_out / w_out (columns >= the device count are ignored, whatever they hold).
    `row_amax` (device int32 [>= M], zero on entry): also maximise, per row, the fp16 bit patterns |y| over the columns whose bit in
    `col_mask` (int32 words, bit n) is clear - the next layer's pre-pass maximum (mixq_gemm_i8_fused_amax; int8, fragment-order
    weights only)."""
    _dev_check(q_x, q_w, x_scale, scale_col, row_amax, col_mask)
    if x_scale.numel() < M or scale_col.numel() < N:
        raise RuntimeError("FusedLinear: x_scale / scale_col are shorter than M / N")
    x_fmt = fmt_of(q_x) if x_packed is None else (FMT_P16X64 if x_packed else FMT_PLAIN)
    w_fmt = fmt_of(q_w) if w_packed is None else (FMT_P16X64 if w_packed else FMT_PLAIN)
    # (ACT_SILU_PAIR: N counts the interleaved gate / up rows, the product has N / 2 columns)
    if act == ACT_SILU_PAIR and (N % 16 or not _is_zero_addend(addend)):
        raise RuntimeError("FusedLinear: ACT_SILU_PAIR needs N % 16 == 0 and takes no addend")
    y = out if out is not None else torch.empty((M, N // 2 if act == ACT_SILU_PAIR else N), dtype=torch.float16, device=q_x.device)
    if n_out and x_out is not None and w_out is not None:
        xop, ldxo = _rows(x_out, "x_out")
        wop, ldwo = _rows(w_out, "w_out")
    else:
        xop = wop = None
        ldxo = ldwo = 0
        n_out = 0
    if act == ACT_SILU_MUL:                  # the addend slot carries the multiplier (silu(gate) * up): never dropped
        if addend is None:
            raise RuntimeError("FusedLinear: ACT_SILU_MUL needs the multiplier in `addend`")
        _dev_check(addend)
        ap, lda = _rows(addend, "addend")
    elif _is_zero_addend(addend):
        ap, lda = None, 0
    else:
        if isinstance(addend, PendingOutlierProduct):
            addend._materialize()
        _dev_check(addend)
        ap, lda = _rows(addend, "addend")
    if row_amax is not None:
        if bit != 8 or row_amax.numel() < M or row_amax.element_size() != 4:
            raise RuntimeError("FusedLinear: row_amax needs an int8 GEMM and a 4-byte buffer of at least M entries")
        _capi.call("mixq_gemm_i8_fused_amax", q_x.data_ptr(), q_w.data_ptr(), x_scale.data_ptr(), scale_col.data_ptr(), xop, ldxo, wop, ldwo,
                   n_out, _ptr(n_out_dev), ap, lda, _ptr(bias), y.data_ptr(), y.stride(0), M, N, K, act, _layout_bits(x_fmt, w_fmt),
                   row_amax.data_ptr(), _ptr(col_mask), _stream())
        return y
    fn = "mixq_gemm_i8_fused" if bit == 8 else "mixq_gemm_i4_fused"
    _capi.call(fn, q_x.data_ptr(), q_w.data_ptr(), x_scale.data_ptr(), scale_col.data_ptr(), xop, ldxo, wop, ldwo, n_out,
               _ptr(n_out_dev), ap, lda, _ptr(bias), y.data_ptr(), y.stride(0), M, N, K, act, _layout_bits(x_fmt, w_fmt), _stream())
    return y


class ForwardPlan:
    """The argument block of mixq_linear_forward (include/mixq_hip.h) for one frozen layer and batch size, built once and re-used:
    a forward then costs three allocations, four pointer updates and ONE foreign call that enqueues the quantise pass and the
    GEMM (the steady state of linear.py:187-193 + :244-285).  Holds references to every tensor whose address it carries."""

    __slots__ = ("args", "ref", "fn", "keep", "M", "N", "K", "ldx", "n", "ldxo", "q_shape", "q_dtype", "qfmt", "device", "lock", "captured", "kept_mask")

    def __init__(self, M, N, K, bit, sigma, ldx, ind_buf, n, n_dev, x_scale, q_w, scale_col, w_out, bias, qfmt, act=ACT_NONE, kept_mask=None):
        import ctypes as C
        _dev_check(x_scale, q_w, scale_col, ind_buf, n_dev, w_out, bias)
        if x_scale.numel() < M or scale_col.numel() < N:
            raise RuntimeError("ForwardPlan: x_scale / scale_col are shorter than M / N")
        n_cap = 0 if ind_buf is None else int(ind_buf.numel())
        if n_cap:
            ind_buf = _check_ind(ind_buf, "ForwardPlan")                # (int32, contiguous - before its address is taken)
        a = _capi.LinearArgs()
        a.ldx, a.M, a.N, a.K, a.bit, a.sigma, a.act = ldx, M, N, K, bit, float(sigma), act
        a.qfmt, a.wfmt = qfmt, fmt_of(q_w)
        a.ind, a.n_cap, a.n_dev = _ptr(ind_buf), n_cap, _ptr(n_dev)
        a.x_scale, a.q_w, a.scale_col, a.bias = x_scale.data_ptr(), q_w.data_ptr(), scale_col.data_ptr(), _ptr(bias)
        self.ldxo = a.ldxo = _pad16(n_cap) if n_cap else 0
        if n_cap:
            wop, ldwo = _rows(w_out, "w_out")
            if ldwo < n_cap or ldwo % 8:
                raise RuntimeError("ForwardPlan: w_out row stride must be >= the outlier capacity and a multiple of 8")
            a.w_out, a.ldwo = wop, ldwo
        a.ldy = N
        self.args, self.ref, self.fn = a, C.byref(a), _capi.load().mixq_linear_forward
        # kept_mask: the layer's kept outlier map of `ind` (include/mixq_hip.h: bits, count, AND-masks): the quantise pass then needs one memory round trip
        if kept_mask is not None:
            _dev_check(kept_mask)
            _check_kept_map(kept_mask, K, "ForwardPlan")
        self.kept_mask = kept_mask if n_cap else None
        self.keep = (ind_buf, n_dev, x_scale, q_w, scale_col, w_out, bias, kept_mask)
        self.M, self.N, self.K, self.ldx, self.n, self.qfmt, self.device = M, N, K, ldx, n, qfmt, x_scale.device
        import threading
        self.lock = threading.Lock()                   # (the argument block is ONE structure: filled and handed over under the lock, see run)
        self.captured = False                          # ran under hipGraph capture: a graph out there replays the addresses this plan carries
        self.q_shape, self.q_dtype = (packed_rows(M) if qfmt else M, _q_row_bytes(K, bit, qfmt)), (torch.int8 if bit == 8 else torch.uint8)

    def run(self, x, row_amax=None, col_mask=None):
        """x: fp16 [M,K] with the row stride the plan was built for.  Returns (y [M,N], q_x, x_out [M,n] or None).
        row_amax (int32 [>= M] on the device): the rows' masked maxima left by the GEMM that produced x (FusedLinear(row_amax=...));
        the quantise pass is then the one-pass known-maximum kernel, which also clears the buffer."""
        dev = self.device
        if col_mask is not None and col_mask is not self.kept_mask:
            _dev_check(col_mask)
            _check_kept_map(col_mask, self.K, "ForwardPlan.run")
        # the block carries x by address: the same checks on every call as the two-call route's QuantFused
        if x.device != dev:
            raise RuntimeError(f"ForwardPlan: x lives on {x.device}, the plan was built for {dev}")
        if x.dtype != torch.float16 or x.dim() != 2 or x.shape[0] != self.M or x.shape[1] != self.K or x.stride(1) != 1 or x.stride(0) != self.ldx:
            raise RuntimeError(f"ForwardPlan: x must be float16 [{self.M}, {self.K}] with row stride {self.ldx} and a contiguous last dimension "
                               f"(got {x.dtype} {tuple(x.shape)} strides {tuple(x.stride())})")
        q = torch.empty(self.q_shape, dtype=self.q_dtype, device=dev)
        y = torch.empty((self.M, self.N), dtype=torch.float16, device=dev)
        xo = torch.empty((self.M, self.ldxo), dtype=torch.float16, device=dev) if self.ldxo else None
        if not self.captured and torch.cuda.is_current_stream_capturing():
            self.captured = True
        a = self.args
        # ONE structure per plan and a foreign call that releases the GIL: two threads (or two streams driven from two threads) running
        # the same frozen layer must not interleave "fill" and "launch" - the C side reads the block before it returns
        with self.lock:
            cm = col_mask if col_mask is not None else self.kept_mask
            a.row_amax, a.col_mask, a.col_mask_words = _ptr(row_amax), _ptr(cm), (0 if cm is None else cm.numel())
            if xo is not None:
                a.x_out = xo.data_ptr()
            a.x, a.q_x, a.y = x.data_ptr(), q.data_ptr(), y.data_ptr()
            rc = self.fn(self.ref, torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise _capi.MixqError("mixq_linear_forward", rc)
        q._mixq_fmt = self.qfmt
        return y, q, (xo[:, :self.n] if xo is not None else None)


# ------------------------------------------------------------------------------------------------------------
# RMSNorm in front of the path (mixquant/modules/fused/norm.py:21-33)
# ------------------------------------------------------------------------------------------------------------
def layernorm_forward_cuda(x, weight, out, eps):
    """mixlib.layernorm_forward_cuda(x, weight, out, eps): out = RMSNorm(x) * weight (FasterTransformer form)."""
    _dev_check(x, weight, out)
    K = x.shape[-1]
    x2, o2 = x.reshape(-1, K), out.reshape(-1, K)
    xp, ldx = _rows(x2, "x")
    op, ldo = _rows(o2, "out")
    _capi.call("mixq_rmsnorm", xp, weight.data_ptr(), op, x2.shape[0], K, ldx, ldo, float(eps), _stream())
    return out


def RMSNormQuantFused(x, weight, out, eps, ind, x_scale, bit, sigma=6.0, flag=None, packed=False, fmt=None, n_dev=None, col_mask=None):
    """RMSNorm fused with the next linear's extract + zero + scale + quantise.  Returns (q_x, x_out_view[M,n]); q_x carries
    its format tag.  `col_mask`: the next layer's kept outlier map of `ind` (as QuantFused)."""
    _dev_check(x, weight, out, x_scale, ind, col_mask)
    fmt = _want_fmt(packed, fmt)
    K = x.shape[-1]
    x2, o2 = x.reshape(-1, K), out.reshape(-1, K)
    xp, ldx = _rows(x2, "x")
    op, ldo = _rows(o2, "out")
    M = x2.shape[0]
    if x_scale.numel() < M:
        raise RuntimeError(f"RMSNormQuantFused: x_scale holds {x_scale.numel()} rows, the batch has {M}")
    n = 0 if ind is None else ind.numel()
    q = torch.empty((packed_rows(M) if fmt else M, _q_row_bytes(K, bit, fmt)),
                    dtype=torch.int8 if bit == 8 else torch.uint8, device=x.device)
    if n:
        ind = _check_ind(ind, "RMSNormQuantFused")
        x_out = torch.empty((M, (n + 15) // 16 * 16), dtype=torch.float16, device=x.device)
        xop, ldxo, ip = x_out.data_ptr(), x_out.stride(0), ind.data_ptr()
    else:
        x_out, xop, ldxo, ip = None, None, 0, None
    if col_mask is not None and n:
        _check_kept_map(col_mask, K, "RMSNormQuantFused")
        _capi.call("mixq_rmsnorm_quant_fused_masked", xp, weight.data_ptr(), op, ip, n, _ptr(n_dev), col_mask.data_ptr(), col_mask.numel(), x_scale.data_ptr(),
                   q.data_ptr(), xop, _ptr(flag), M, K, ldx, ldo, ldxo, float(eps), bit, float(sigma), fmt, _stream())
    else:
        _capi.call("mixq_rmsnorm_quant_fused", xp, weight.data_ptr(), op, ip, n, _ptr(n_dev), x_scale.data_ptr(), q.data_ptr(), xop,
                   _ptr(flag), M, K, ldx, ldo, ldxo, float(eps), bit, float(sigma), fmt, _stream())
    return set_fmt(q, fmt), (x_out[:, :n] if n else None)


def layernorm_forward_cuda_extract_outliers(x, weight, out, eps, ind, x_scale):
    """mixlib.layernorm_forward_cuda_extract_outliers(x, w, out, eps, ind, x_scale) -> (X_out, q_x) (norm.py:25-28)."""
    q, xo = RMSNormQuantFused(x, weight, out, eps, ind, x_scale, 8)
    if xo is None:
        xo = torch.empty((q.shape[0], 0), dtype=torch.float16, device=x.device)
    return xo, q


def layernorm_forward_cuda_extract_outliers_int4(x, weight, out, eps, ind, x_scale):
    """mixlib.layernorm_forward_cuda_extract_outliers_int4 (norm.py:30-33)."""
    q, xo = RMSNormQuantFused(x, weight, out, eps, ind, x_scale, 4)
    if xo is None:
        xo = torch.empty((q.shape[0], 0), dtype=torch.float16, device=x.device)
    return xo, q


# ------------------------------------------------------------------------------------------------------------
# weight-only W8A16 (the reference reaches it through EETQ, linear.py:9,102-106,178-184)
# ------------------------------------------------------------------------------------------------------------
def PackW8A16(q_weight_kn):
    """Re-tile the checkpoint's int8 [K,N] weight for mixq_gemm_w8a16 (once per layer) -> uint8 [round16(N) * K]."""
    _dev_check(q_weight_kn)
    if q_weight_kn.dtype != torch.int8 or q_weight_kn.dim() != 2 or not q_weight_kn.is_contiguous():
        raise RuntimeError("PackW8A16: expected a contiguous int8 [K,N] tensor")
    K, N = q_weight_kn.shape
    out = torch.empty(((N + 15) // 16 * 16) * K, dtype=torch.uint8, device=q_weight_kn.device)
    _capi.call("mixq_pack_w8a16", q_weight_kn.data_ptr(), out.data_ptr(), K, N, _stream())
    return out


def W8A16Linear(x, w_packed, scale_col, bias, N, K, out=None):
    """y = x @ (q * scale_col) + bias with the weights from PackW8A16.  x fp16 [M,K] (row stride % 8 == 0)."""
    _dev_check(x, w_packed, scale_col, bias)
    if x.dtype != torch.float16 or scale_col.dtype != torch.float16:
        raise RuntimeError("W8A16Linear: x and scale_col must be float16")
    x2 = x.reshape(-1, x.shape[-1])
    if x2.shape[1] != K or scale_col.numel() != N:
        raise RuntimeError("W8A16Linear: shape mismatch")
    if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    M = x2.shape[0]
    y = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=x.device)
    _capi.call("mixq_gemm_w8a16", x2.data_ptr(), x2.stride(0) if M else K, w_packed.data_ptr(), scale_col.data_ptr(), _ptr(bias),
               y.data_ptr(), y.stride(0) if M else N, M, N, K, _stream())
    return y
