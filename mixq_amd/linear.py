"""MixLinear_GEMM — MixQ's mixed-precision quantized Linear (W8A8O16 / W4A4O16) for MI355X.

API mirror of /root/reference/mixquant/modules/linear.py (constructor :27-28, from_linear :90-91, forward :165,
forward_without_preconditionFusedSilu :292, FindOutliers :157, helpers two_compl/pack_to_i4/unpack_int8_to_int4 :12-22;
buffer names/shapes/dtypes :39-65), so it drops into the reference's Llama modules (`W_pack(x)`, `o_proj(x, None, True)`,
`up_proj_(x, cache)`, `gate_proj_.forward_without_preconditionFusedSilu(x, cache)`) unchanged.

What is different underneath:
  * the k2+k1 pre-pass (extract outliers, zero them, row scale, quantise) is ONE kernel that also evaluates the
    misprediction predicate of linear.py:201 on device;
  * the cuBLAS outlier GEMM + M x N addend round trip (linear.py:248-256) is fused into the int8 MFMA GEMM as fp16
    MFMA tail iterations; bias is added in the same epilogue;
  * outlier detection (torch.unique(torch.where(...)), linear.py:160) is a column-flag kernel + compaction;
  * from_linear does NOT destroy the caller's `linear.weight` (the reference divides it in place when it is already
    on the device, linear.py:116-117); the produced q_weight/scale_col are identical.
All compute goes through `mixq_amd.mixlib` -> libmixq_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from . import mixlib as _hip_mixlib
from .config import MixqConfig
from ._capi import ACT_NONE, ACT_SILU, ACT_SILU_MUL, FMT_PLAIN, FMT_P16X64, FMT_F16X64, FMT_F6X128, FMT_R6X128

# The kernel backend: always the HIP module.  (The host-side tests that run without a GPU rebind this module attribute to an oracle-backed
# stand-in themselves - tests/conftest.py: swap_backend - the product carries no switch for it.)
_backend = _hip_mixlib


def _capturing():
    """Is the current HIP stream being captured into a graph?  (torch.cuda.is_current_stream_capturing raises without a device.)"""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _fmt_of(t):
    return getattr(t, "_mixq_fmt", FMT_PLAIN)


def two_compl(x: Tensor, bits: int) -> Tensor:
    """Signed -> unsigned two's-complement field of `bits` bits (linear.py:12-13)."""
    return torch.where(x < 0, x + (1 << bits), x)


def pack_to_i4(X: Tensor) -> Tensor:
    """int8 values in [-8,7], shape [R,K] -> uint8 [R,K/2]; low nibble = even column (linear.py:14-18)."""
    u = two_compl(X.to(torch.int8), 4).to(torch.uint8)
    return u[:, 0::2] | (u[:, 1::2] << 4)


def unpack_int8_to_int4(weight: Tensor, ind: Tensor) -> Tensor:
    """Sign-extended int4 weight columns `ind` as fp16 [N,n] (linear.py:20-22)."""
    assert weight.dim() == 2
    return _backend.unpack_int4_to_fp16(weight, ind)


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


class _ColStore:
    """Growable [rows, cap] fp16 storage whose first n columns are live; row stride stays a multiple of 16 elements
    so the GEMM's fp16 MFMA tail can read 16-byte fragments (include/mixq_hip.h: ldxo/ldwo)."""

    def __init__(self, rows, device, init: Tensor | None = None):
        self.rows, self.device = rows, device
        self.n = 0
        self.buf = None
        if init is not None and init.numel():
            self.append(init)

    def append(self, cols: Tensor):
        k = cols.shape[1]
        if k == 0:
            return
        need = _pad16(self.n + k)
        if self.buf is None or self.buf.shape[1] < need:
            nb = torch.zeros((self.rows, max(need, 32)), dtype=torch.float16, device=self.device)
            if self.buf is not None and self.n:
                nb[:, :self.n] = self.buf[:, :self.n]
            self.buf = nb
        self.buf[:, self.n:self.n + k] = cols
        self.n += k

    def view(self):
        if self.buf is None:
            return None
        return self.buf[:, :self.n]


def _gemm_ready(t: Tensor | None) -> Tensor | None:
    """Return `t` (a [R,n] fp16 matrix) in a layout the GEMM tail accepts: row stride >= pad16(n), multiple of 8."""
    if t is None or t.shape[1] == 0:
        return None
    n = t.shape[1]
    if t.stride(1) == 1 and t.stride(0) >= _pad16(n) and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0:
        return t
    buf = torch.zeros((t.shape[0], _pad16(n)), dtype=torch.float16, device=t.device)
    buf[:, :n] = t
    return buf[:, :n]


class _Derived:
    """Everything a layer DERIVES from its state (weights, `ind`, `weight_cache`, device) and keeps between forwards, in one place:

      wpk / wpk_small   the packed weight image(s) the GEMM streams (after compaction `wpk` is the layer's ONLY copy of the weights)
      wo_ready          `weight_cache` in the GEMM tail's padded layout
      ind_buf, n_dev    `ind` padded to a capacity + the live count in device memory
      cmask             bit-per-input-column mask of `ind` (a producer's row-maximum side output)
      amax_buf          int32 row maxima left by the GEMM that produced this layer's input; allocated ONCE at the cache's row capacity
      plan(s)           kept argument blocks of the one-call forward (raw device addresses of everything above)
      joint             set by MixLlamaMLP (fused.py) once this layer's weights live in the block's joint gate / up image: the layer then
                        keeps NO image of its own (`wpk` None) and re-creates rows from that image on the cold paths

    Each derived item carries the identity (`id`, `data_ptr`, `_version`) of what it was made from and is rebuilt when that changes;
    `invalidate` is the single place that drops them - and every call of it drops the kept argument blocks, which hold addresses of
    all the others."""

    __slots__ = ("wpk", "wpk_key", "wpk_small", "wpk_small_key", "wo_ready", "wo_key", "ind_buf", "ind_key", "n_dev", "n_dev_host",
                 "cmask", "cmask_key", "amax_buf", "amax_dirty", "plan", "plan_key", "plans", "retired", "joint")

    def __init__(self):
        self.wpk = self.wpk_key = self.wpk_small = self.wpk_small_key = None
        self.wo_ready = self.wo_key = self.ind_buf = self.ind_key = None
        self.n_dev, self.n_dev_host = None, -1
        self.cmask = self.cmask_key = None
        self.amax_buf, self.amax_dirty = None, False
        self.plan = self.plan_key = None
        self.plans = {}
        self.retired = []                     # superseded device buffers a captured graph may still address: kept alive, never re-used
        self.joint = None

    def invalidate(self, weights=False, outliers=False, device=False):
        """weights: the weight bytes or their layout changed (checkpoint load into a compacted layer); outliers: `ind` / `weight_cache`
        were replaced; device: the module moved.  The kept argument blocks go in every case."""
        self.plan = self.plan_key = None
        self.plans = {}
        if weights or device:
            if weights:
                self.wpk = None
                self.joint = None             # (new weight bytes: the block re-builds its joint image from them)
            self.wpk_key = None
            self.wpk_small = self.wpk_small_key = None
        if outliers or device:
            self.wo_ready = self.wo_key = None
            self.ind_buf = self.ind_key = None
            self.cmask = self.cmask_key = None
        if device:
            self.n_dev, self.n_dev_host = None, -1
            self.amax_buf, self.amax_dirty = None, False
            self.retired = []


def _derived_attr(name):
    """The pre-round-4 attribute names (`layer._wpk`, `layer._plan`, ...) as views of the record: tests and tools read them."""
    return property(lambda self: getattr(self._d, name), lambda self, v: setattr(self._d, name, v))


def kept_outlier_map(ind, K):
    """The kept outlier map of `ind` over K input columns as int32 words on ind's device (layout: MixLinear_GEMM._col_mask)."""
    dev = ind.device
    n = int(ind.shape[0])
    words = (K + 31) // 32
    bits = torch.zeros(words * 32, dtype=torch.int64, device=dev)
    bits[ind.long()] = 1
    w = (bits.view(words, 32) << torch.arange(32, device=dev, dtype=torch.int64)).sum(dim=1)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)           # the same 32 bits as a signed word
    head = torch.zeros(((words + 1 + 3) // 4) * 4, dtype=torch.int32, device=dev)
    head[:words] = w
    head[words] = n
    kp = ((K + 7) // 8) * 8
    keep = torch.full((kp,), 0xffff, dtype=torch.int64, device=dev)          # per-column AND-mask: 0x0000 for an outlier column
    keep[ind.long()] = 0
    pw = keep.view(kp // 2, 2)
    pw = pw[:, 0] | (pw[:, 1] << 16)
    pw = torch.where(pw >= 2 ** 31, pw - 2 ** 32, pw).to(torch.int32)
    return torch.cat([head, pw])


class MixLinear_GEMM(nn.Module):
    _wpk, _wpk_key = _derived_attr("wpk"), _derived_attr("wpk_key")
    _wpk_small, _wpk_small_key = _derived_attr("wpk_small"), _derived_attr("wpk_small_key")
    _wo_ready, _wo_key = _derived_attr("wo_ready"), _derived_attr("wo_key")
    _ind_buf, _ind_key = _derived_attr("ind_buf"), _derived_attr("ind_key")
    _n_dev, _n_dev_host = _derived_attr("n_dev"), _derived_attr("n_dev_host")
    _cmask, _cmask_key = _derived_attr("cmask"), _derived_attr("cmask_key")
    _amax_buf, _amax_dirty = _derived_attr("amax_buf"), _derived_attr("amax_dirty")
    _plan, _plan_key, _plans = _derived_attr("plan"), _derived_attr("plan_key"), _derived_attr("plans")

    def __init__(self, in_features, out_features, bias, dev, bit, weight_only=False, cache=None, fp_features_num=128,
                 name=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.bit = bit
        if bit not in (4, 8):
            raise ValueError("MixLinear_GEMM: bit must be 4 or 8")
        # the kernels' shape contract (include/mixq_hip.h): said here, where the layer is built, not by the first forward's MIXQ_ESHAPE
        if not weight_only and (in_features % (64 if bit == 8 else 128) or out_features % 4):
            raise ValueError(f"MixLinear_GEMM: W{bit}A{bit} needs in_features % {64 if bit == 8 else 128} == 0 and out_features % 4 == 0 "
                             f"(got {in_features} -> {out_features})")

        if weight_only is False:
            self.register_buffer("scale_col", torch.empty((1, out_features), dtype=torch.float16, device=dev))
            if bit == 8:
                self.ind = torch.zeros((0,), dtype=torch.int32, device=dev)
                self.register_buffer("q_weight", torch.empty((out_features, in_features), dtype=torch.int8, device=dev))
                self.weight_cache = None
            else:
                self.fp_features_num = fp_features_num
                self.register_buffer("q_weight", torch.empty((out_features, in_features // 2), dtype=torch.uint8, device=dev))
                self.register_buffer("weight_cache", torch.empty((out_features, fp_features_num), dtype=torch.float16, device=dev))
                self.register_buffer("ind", torch.empty((fp_features_num,), dtype=torch.int32, device=dev))
        else:
            self.register_buffer("q_weight", torch.empty((in_features, out_features), dtype=torch.int8, device=dev))
            self.register_buffer("scale_col", torch.empty((out_features,), dtype=torch.float16, device=dev))

        if bias:
            self.register_buffer("bias", torch.empty((out_features,), dtype=torch.float16, device=dev))
        else:
            self.bias = None
        self.cnt = 0
        self.forward_without_precondition_len = -1
        if bit == 4:
            self.forward_without_precondition_len = fp_features_num

        self.cache = cache
        # the switches of the model this layer belongs to (config.py): the cache's, shared by every layer built with it; a layer without a
        # cache gets its own.  A plain attribute (not a buffer, not in the state_dict).
        object.__setattr__(self, "config", getattr(cache, "config", None) or MixqConfig())
        self.weight_only = weight_only
        self.add_outliers = True
        if cache is not None:
            self.sigma = torch.ones((1, 1), dtype=torch.float16, device=dev)
            self.sigma[0] = cache.sigma
            self._sigma_f = float(cache.sigma.float().cpu().item())
        self.arch = "gfx950"
        self.name = name
        self._wstore = None          # _ColStore behind weight_cache once outliers were appended online
        self._silu_calls = 0
        object.__setattr__(self, "_d", _Derived())   # every derived, rebuildable item (packed images, padded operands, kept argument blocks)

    # ------------------------------------------------------------------------------------------------------
    @classmethod
    def from_linear(cls, linear, bit, weight_only=False, init_only=False, cache=None, layer_scales=None, dev="cuda",
                    name=None, fp_features_num=128):
        q = cls(linear.in_features, linear.out_features, linear.bias is not None, dev, bit=bit, weight_only=weight_only,
                cache=cache, name=name, fp_features_num=fp_features_num)
        if init_only is True:
            return q
        if weight_only is True:
            # linear.py:102-106: EETQ's per-column symmetric int8 of W^T [K,N] (restated in mixq_amd/eetq.py)
            from . import eetq
            int8_weight, scales = eetq.quant_weights(torch.t(linear.weight.data).contiguous().cpu(), torch.int8, False)
            q.q_weight.copy_(int8_weight)
            q.scale_col.copy_(scales.half())
            if linear.bias is not None:
                q.bias.copy_(linear.bias.data.half())
            return q

        W = linear.weight.data.to(dev)                      # a copy when it moves; cloned below before any in-place op
        N = linear.out_features
        if bit == 8:
            # linear.py:113-119: scale = fp16(rowabsmax/127); q = round(W / scale) in W's dtype, cast to int8
            scale = (torch.max(torch.abs(W), dim=1)[0].unsqueeze(1) / 127).to(torch.float16).reshape((1, N))
            q.scale_col.copy_(scale)
            tmp = W.clone()
            tmp /= q.scale_col.T
            q.q_weight.copy_(tmp.round().to(torch.int8))
        else:
            # linear.py:123-143: keep the `fp_features_num` input channels with the largest activation scale in fp16,
            # zero them in W, scale = rowabsmax/10, clamp to [-8,7], nibble-pack.
            assert layer_scales is not None
            ind = torch.sort(layer_scales)[1][-q.fp_features_num:]
            linear.ind = ind
            ind_d = ind.to(dev)
            q.weight_cache.copy_(W[:, ind_d].to(W.dtype))
            tmp = W.clone()
            tmp[:, ind_d] = 0
            scale = (torch.max(torch.abs(tmp), dim=1)[0].unsqueeze(1) / 10).to(torch.float16).reshape((1, N))
            q.scale_col.copy_(scale)
            tmp /= q.scale_col.T
            tmp = torch.clamp(tmp.round(), -8, 7)
            q.q_weight.copy_(pack_to_i4(tmp.to(torch.int8).cpu()).to(dev))
            q.ind.copy_(ind_d.to(torch.int32))
        if linear.bias is not None:
            q.bias.copy_(linear.bias.data.half())
        return q

    # ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def FindOutliers(self, Activation):
        """Sorted distinct column ids with any |x| > sigma, int32 (linear.py:157-161).  One host read of the count."""
        cache = self.cache
        scratch = cache.scratch(Activation.shape[1]) if hasattr(cache, "scratch") else None
        ind_buf, count = _backend.DetectOutlierCols(Activation, self._sigma_f, scratch)
        n = int(count.item())
        return ind_buf[:n].clone()

    def _append_outliers(self, cache, inputs, ind):
        """linear.py:205-221: pull the new columns out of x, dequantise the matching weight columns, append both."""
        activation_outliers = _backend.ExtractOutliersAndSetToZeros(ind, inputs)
        weight_cache = _backend.DequantWeightCols(self.q_weight, self.scale_col, ind, self.bit)
        if self._wstore is None:
            self._wstore = _ColStore(self.out_features, weight_cache.device,
                                     self.weight_cache if (self.weight_cache is not None and self.ind.shape[0]) else None)
        xs = _ColStore(inputs.shape[0], inputs.device,
                       cache.activation_outliers if (self.ind.shape[0] and cache.activation_outliers is not None) else None)
        xs.append(activation_outliers)
        self._wstore.append(weight_cache)
        cache.activation_outliers = xs.view() if xs.n else activation_outliers
        self.weight_cache = self._wstore.view() if self._wstore.n else weight_cache
        self.ind = torch.hstack((self.ind, ind))
        cache.ind = self.ind

    # ---- weights: one packed image; the plain matrix only while the outlier search may still need its columns ----------
    def __getattr__(self, name):
        if name == "q_weight":
            d = self.__dict__
            bufs = d.get("_buffers")
            if bufs is not None and "q_weight" in bufs and bufs["q_weight"] is None and "_d" in d and \
                    (d["_d"].wpk is not None or d["_d"].joint is not None):
                return self._plain_weight()
        return super().__getattr__(name)

    def _plain_weight(self):
        """The reference-layout q_weight re-created from the packed image (cold paths: state_dict, QKV fusion, new outlier
        columns after compaction).  A fresh tensor every time: writes to it do not reach the layer - load_state_dict does."""
        if self.weight_only:
            raise RuntimeError("weight-only layers keep their plain q_weight")
        if self._wpk is None and self._d.joint is not None:
            return self._d.joint.plain_rows()                            # this layer's rows of the MLP block's joint gate / up image
        if not self._wpk.is_cuda:
            # the module was moved to the host (model.cpu() before saving): the re-tiling is pure index arithmetic, done in torch
            return _unpack_host(self._wpk, self.out_features, _fmt_of(self._wpk))
        return _backend.UnpackOperand(self._wpk, self.out_features)

    def compact_weights_(self):
        """Drop the plain [N,K] copy of q_weight, keeping only the packed image (half the weight memory of the first round)."""
        if self.weight_only or self._buffers.get("q_weight") is None:
            return self
        if self._d.joint is not None and self._wpk is None:
            self._buffers["q_weight"] = None                             # (the joint image holds these rows)
            return self
        if self._packed_weight() is not None:
            self._buffers["q_weight"] = None
        return self

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if "q_weight" in self._buffers and self._buffers["q_weight"] is None and (self._wpk is not None or self._d.joint is not None):
            destination[prefix + "q_weight"] = self._plain_weight()      # the reference's on-disk layout (base.py:78-119)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        compacted = "q_weight" in self._buffers and self._buffers["q_weight"] is None and (self._wpk is not None or self._d.joint is not None)
        if compacted:
            key = prefix + "q_weight"
            if key not in state_dict:
                missing_keys.append(key)                                 # nn.Module skips None buffers: report it ourselves
            else:
                src = state_dict[key]
                KB = self.in_features if self.bit == 8 else self.in_features // 2
                want_dtype = torch.int8 if self.bit == 8 else torch.uint8
                if tuple(src.shape) != (self.out_features, KB) or src.dtype != want_dtype:
                    # keep the packed image: a mismatching tensor must not leave the layer with uninitialised weights
                    error_msgs.append(f"size mismatch for {key}: copying a param with shape {tuple(src.shape)} ({src.dtype}) from checkpoint, "
                                      f"the shape in current model is {(self.out_features, KB)} ({want_dtype}).")
                    state_dict = {k: v for k, v in state_dict.items() if k != key}
                else:
                    self._buffers["q_weight"] = torch.empty(tuple(src.shape), dtype=src.dtype, device=self.scale_col.device)
                    self._d.invalidate(weights=True)                     # re-packed on the next forward; kept plans pin the old images
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def _apply(self, fn, *args, **kwargs):
        """nn.Module.to / .cuda / .half walk parameters and buffers only; after compaction the weights live in `_wpk` (a plain
        attribute), so the packed image follows the module here and everything derived from the old tensors is rebuilt."""
        d = self._d
        captured = [pl for pl in d.plans.values() if getattr(pl, "captured", False)]
        if captured and not getattr(self, "allow_move_after_capture", False):
            here = captured[0].device
            if fn(torch.empty(0, device=here)).device != here:
                # a hipGraph captured through this layer replays raw addresses of the weight image, the outlier operands and the device
                # count ON THIS DEVICE; moving the module frees them, and nothing can stop the graph from being replayed afterwards.
                raise RuntimeError("MixLinear_GEMM: this layer ran under hipGraph capture; moving it to another device would leave that graph "
                                   "with dangling addresses.  Drop the graph first, then set layer.allow_move_after_capture = True (re-capture "
                                   "after the move).")
        def idents():
            own = [self.__dict__.get("weight_cache"), self.__dict__.get("ind"), d.wpk]
            return [id(t) for t in list(self._buffers.values()) + own]
        before = idents()
        out = super()._apply(fn, *args, **kwargs)
        # fn is applied ONCE per tensor (a real move copies: the 45 MB image must not be copied to find out whether anything moves - ADVICE r05)
        wc, ind = self.__dict__.get("weight_cache"), self.__dict__.get("ind")
        moved_wpk = fn(d.wpk) if d.wpk is not None else None
        moved_wc = fn(wc) if isinstance(wc, Tensor) else wc
        moved_ind = fn(ind) if isinstance(ind, Tensor) else ind
        if moved_wpk is d.wpk and moved_wc is wc and moved_ind is ind and idents() == before:
            # an _apply that changes nothing (model.half() on an fp16 layer): the derived buffers - among them the row-maximum
            # hand-over buffer and the retired images a captured graph still writes to and reads - stay as they are
            return out
        if d.wpk is not None and moved_wpk is not d.wpk:
            _backend.set_fmt(moved_wpk, _fmt_of(d.wpk))
            d.wpk = moved_wpk
        # everything else derived is rebuilt where the module now lives: the image is re-packed from q_weight when that buffer still
        # exists, the small-batch image when such a batch arrives, and the kept argument blocks - which pin the OLD device's tensors and
        # which a graph captured before the move may still replay (tests/test_gpu_round4.py) - are dropped
        d.invalidate(device=True)
        if isinstance(wc, Tensor):
            self.weight_cache = moved_wc                                 # 8-bit layers keep it as a plain attribute (linear.py:42)
        if isinstance(ind, Tensor):
            self.ind = moved_ind
        self._wstore = None
        return out

    def x_fmt(self, M=None):
        """Layout this layer wants its quantised activation in for a batch of M rows (what a fused norm in front of it should emit)."""
        if self._d.joint is not None and self._wpk is None and self._buffers.get("q_weight") is None and not self._small_batch_image(M):
            return FMT_P16X64 if self.bit == 8 else FMT_R6X128           # (the joint gate / up image: fragment-order int8 / FP6 codes, as below)
        wpk = self._packed_weight(M)
        if wpk is None:
            return FMT_PLAIN
        # fragment-order int8 weights go with P16X64 activations; FP6-coded weights with FP6-coded, row-contiguous activations
        return FMT_R6X128 if _fmt_of(wpk) == FMT_F6X128 else FMT_P16X64

    def _small_batch_image(self, M):
        return self.bit == 4 and M is not None and 0 < M <= self.config.small_batch_m4 and self.config.pack_fmt4 == FMT_F6X128 and not self.weight_only

    def _packed_small(self):
        """The nibble (P16X64) image of a 4-bit layer, for small batches; made from the plain matrix while it exists, from the FP6 image after."""
        qw = self._buffers.get("q_weight")
        main = self._packed_weight()
        if main is None:
            return None
        # keyed on the MAIN image (which itself follows q_weight while that buffer exists): compaction drops q_weight but not the main
        # image, so the nibble image - an address kept argument blocks and captured graphs hold - survives it
        key = (id(main), main._version)
        if self._wpk_small is None or self._wpk_small_key != key:
            plain = qw if qw is not None else self._plain_weight()
            if self._wpk_small is not None:
                self._d.retired.append(self._wpk_small)
            self._wpk_small, self._wpk_small_key = _backend.PackOperand(plain, FMT_P16X64), key
        return self._wpk_small

    def _packed_weight(self, M=None):
        """q_weight in the tile-major layout the GEMM streams fastest for a batch of M rows (include/mixq_hip.h); rebuilt when the buffer is
        replaced or rewritten (checkpoint load)."""
        if self._small_batch_image(M):
            return self._packed_small()
        qw = self._buffers.get("q_weight")
        if qw is None:
            if self._wpk is None and self._d.joint is not None:
                # the layer is used on its own again (its MLP block took the joint route before): its own image, from the joint one
                self._wpk = _backend.PackOperand(self._plain_weight(), self.config.pack_fmt if self.bit == 8 else self.config.pack_fmt4)
            return self._wpk                                             # compacted: the packed image is all there is
        if qw.shape[1] % 64:
            return None                                                  # (plain operands: the LDS-staged kernel's own loads)
        # W4A4: FP6 codes for the FP6 matrix pipe (config.pack_fmt4); as nibbles it stays with the LDS-staged kernel (P16X64 weights), whose
        # nibble expansion hides behind 32-cycle MFMAs, not behind the 16-cycle ones of the weights-in-registers kernel (28.0 vs 32.4 us)
        fmt = self.config.pack_fmt if self.bit == 8 else self.config.pack_fmt4
        key = (id(qw), qw.data_ptr(), qw._version, fmt)
        if self._wpk is None or self._wpk_key != key:
            self._wpk = _backend.PackOperand(qw, fmt)
            self._wpk_key = key
        return self._wpk

    # ---- outlier bookkeeping the kernels see: `ind` padded to a capacity, the live count in device memory ---------------
    def _ind_dev(self):
        """(ind buffer of capacity pad16(n), device count) for n = len(self.ind) > 0; (None, None) without outliers."""
        n = int(self.ind.shape[0])
        if n == 0:
            return None, None
        ind = self.ind
        key = (id(ind), ind.data_ptr(), ind._version, n)
        if self._ind_buf is None or self._ind_key != key:
            buf = torch.zeros((_pad16(n),), dtype=torch.int32, device=ind.device)
            buf[:n] = ind
            self._ind_buf, self._ind_key = buf, key
        buf = self._ind_buf
        if self._n_dev is None or self._n_dev.device != ind.device:
            self._n_dev = torch.zeros((1,), dtype=torch.int32, device=ind.device)
            self._n_dev_host = -1
        if self._n_dev_host != n:
            self._n_dev.fill_(n)                                         # a device-side write: no host sync
            self._n_dev_host = n
        return buf[:_pad16(n)], self._n_dev

    # ---- this layer's pre-pass maximum as a side output of the GEMM that produces its input (fused/mlp.py:57-70) --------
    def _col_mask(self):
        """This layer's KEPT OUTLIER MAP (include/mixq_hip.h, mixq_quant_fused_masked), int32 words: [ceil(K / 32) words: bit k set <=> input
        column k is one of this layer's outlier columns][ONE word: the number of columns marked - what the quantise passes check a kept map
        against][pad to a multiple of 4 words][K uint16 AND-masks, two per word: 0x0000 for an outlier column, 0xffff elsewhere].  None without outliers.
        A producing GEMM reads only the bits (its row-maximum side output skips the marked columns)."""
        n = int(self.ind.shape[0])
        if n == 0:
            return None
        ind = self.ind
        key = (id(ind), ind.data_ptr(), ind._version, n)
        if self._cmask is None or self._cmask_key != key:
            self._cmask, self._cmask_key = kept_outlier_map(ind, self.in_features), key
        return self._cmask

    def amax_target(self, M, device):
        """(row_amax buffer, column mask) a producing GEMM should fill for this layer's next forward of M rows, or None when this
        layer cannot use it (outlier search still running, weight-only).  The buffer is zero when handed out."""
        if self.weight_only or self.add_outliers or not self.config.one_call_forward:
            return None
        d = self._d
        # ONE buffer for the layer's lifetime on a device, sized for the largest batch the cache admits (x_scale has one row per
        # token, Cache.py:8): a hipGraph captured at a small batch has this address baked into the producer's atomicMax and the
        # quantiser's read-and-clear, so a later, larger eager batch must find the SAME buffer, not a re-allocation
        cap = max(int(self.cache.x_scale.numel()), 16) if self.cache is not None else max(M, 16)
        if M > cap:
            return None                                      # (more rows than the cache holds: that forward fails on x_scale anyway)
        if d.amax_buf is None or d.amax_buf.device != torch.device(device) or d.amax_buf.numel() < cap:
            if d.amax_buf is not None:
                d.retired.append(d.amax_buf)                 # (a grown cache / another device: captured graphs may still address the old one)
            d.amax_buf = torch.zeros((cap,), dtype=torch.int32, device=device)
            d.amax_dirty = False
        if self._amax_dirty:
            self._amax_buf.zero_()                           # a producer ran and nobody consumed: start clean
        self._amax_dirty = True
        return self._amax_buf, self._col_mask()

    def _gemm(self, cache, M, act, addend=None, row_amax=None, col_mask=None):
        n = int(self.ind.shape[0])
        xo = wo = n_dev = None
        cap = 0
        if n:
            xo = cache.activation_outliers
            if xo is None or xo.shape[1] != n:
                raise RuntimeError("MixLinear_GEMM: outlier operands do not match `ind`")
            xo = _gemm_ready(xo)
            # weight_cache is static between outlier appends: re-pad it (e.g. a [N,129] buffer loaded from a checkpoint)
            # once, not on every forward
            wc = self.weight_cache
            key = (id(wc), wc.data_ptr(), wc._version, tuple(wc.shape), wc.stride(0))
            if self._wo_key != key:
                self._wo_ready, self._wo_key = _gemm_ready(wc), key
            wo = self._wo_ready
            if wo is None or wo.shape[1] != n:
                raise RuntimeError("MixLinear_GEMM: outlier operands do not match `ind`")
            # capacity = the padded width both operands really have; the count itself is read from device memory
            cap = min(_pad16(n), xo.stride(0), wo.stride(0))
            if getattr(cache, "n_dev", None) is not None and self._n_dev is cache.n_dev and cap > n:
                n_dev, n_cap = self._n_dev, cap
            else:
                n_dev, n_cap = None, n
        wpk = self._packed_weight(M)
        qx = cache.q_xcache
        w = wpk if wpk is not None else self.q_weight
        want = self.x_fmt(M)
        if _fmt_of(qx) != want:
            # the producer of q_xcache (e.g. the reference's own fused norm through the mixlib shim) used another layout
            if _fmt_of(qx) != FMT_PLAIN:
                qx = _backend.UnpackOperand(qx, M)
            if want != FMT_PLAIN:
                qx = _backend.PackOperand(qx[:M].contiguous(), want)
        extra = {} if row_amax is None else {"row_amax": row_amax, "col_mask": col_mask}
        if n:
            return _backend.FusedLinear(qx, w, cache.x_scale, self.scale_col, _wide(xo, n_cap), _wide(wo, n_cap), n_cap, self.bias, M, self.out_features, self.in_features, bit=self.bit,
                                        act=act, addend=addend, n_out_dev=n_dev, **extra)
        return _backend.FusedLinear(qx, w, cache.x_scale, self.scale_col, None, None, 0, self.bias, M, self.out_features,
                                    self.in_features, bit=self.bit, act=act, addend=addend, **extra)

    def __getstate__(self):
        """copy.deepcopy / pickle of a frozen layer: the kept argument block of the one-call forward is a ctypes structure full of device
        pointers - neither copyable nor meaningful in the copy, which builds its own on its first frozen forward."""
        state = dict(self.__dict__)
        d = _Derived()
        for k in _Derived.__slots__:
            setattr(d, k, getattr(self._d, k))
        d.plan, d.plan_key, d.plans, d.retired = None, None, {}, []
        if d.joint is not None:                              # (the copy stands alone: its own image instead of a reference into the MLP block's)
            if d.wpk is None and self._buffers.get("q_weight") is None:
                d.wpk = _backend.PackOperand(self._plain_weight(), self.config.pack_fmt if self.bit == 8 else self.config.pack_fmt4)
            d.joint = None
        state["_d"] = d
        return state

    # ---- frozen steady state: the whole forward behind ONE foreign call (include/mixq_hip.h: mixq_linear_forward) ------
    def _frozen_key(self, cache, inputs, M):
        """Identity of everything the kept argument block carries an address of: a replaced tensor (new id) or an in-place rewrite
        of a tensor this layer keeps a derived copy of (`ind`, `weight_cache`, `q_weight`: their versions) invalidates the plan."""
        d = self.__dict__
        b = d["_buffers"]
        ind = d.get("ind")
        if ind is None:
            ind = b.get("ind")
        wc = d.get("weight_cache")
        if wc is None:
            wc = b.get("weight_cache")
        qw = b.get("q_weight")
        dd = d["_d"]
        return (M, inputs.stride(0), inputs.device, id(cache), id(cache.x_scale), id(ind), ind._version, id(dd.wpk), id(dd.wpk_small), id(qw),
                -1 if qw is None else qw._version, id(wc), -1 if wc is None else wc._version, id(b.get("bias", d.get("bias"))),
                id(b.get("scale_col"))) + d["config"].key()

    def _build_plan(self, cache, inputs, M):
        if M == 0 or inputs.dtype != torch.float16 or inputs.stride(1) != 1 or inputs.stride(0) % 8:
            return None
        wpk = self._packed_weight(M)
        if wpk is None:
            return None                                      # K % 64: plain operands, the two-call route serves them
        n = int(self.ind.shape[0])
        ind_buf, n_dev = self._ind_dev()
        wo = None
        if n:
            wc = self.weight_cache
            key = (id(wc), wc.data_ptr(), wc._version, tuple(wc.shape), wc.stride(0))
            if self._wo_key != key:
                self._wo_ready, self._wo_key = _gemm_ready(wc), key
            wo = self._wo_ready
            if wo is None or wo.shape[1] != n or wo.stride(0) < _pad16(n):
                return None
            wo = _wide(wo, _pad16(n))
        # (frozen: `ind` never changes again, so the quantise pass gets the kept bit mask of its columns instead of building one per launch)
        return _backend.ForwardPlan(M, self.out_features, self.in_features, self.bit, self._sigma_f, inputs.stride(0), ind_buf, n, n_dev,
                                    cache.x_scale, wpk, self.scale_col, wo, self.bias, self.x_fmt(M), kept_mask=self._col_mask() if n else None)

    # ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, cache=None, unfused=False):
        if cache is None:
            cache = self.cache
        cache.shape = x.shape[:-1] + (self.out_features,)
        inputs = x.reshape(-1, x.shape[-1])
        M = inputs.shape[0]
        if unfused and not self.add_outliers and self.weight_only is False and self.config.one_call_forward:
            # prediction frozen: extract + quantise + GEMM enqueued by one C call on a kept argument block; bit-identical to the
            # route below (tests/test_gpu_round3.py::test_one_call_forward_is_bit_identical)
            # The argument block takes x by ADDRESS: what the two-call route checked in QuantFused is checked here on every call - a
            # kept block must not be replayed on a bf16 / fp32 tensor, another column count or a strided last dimension
            if inputs.dtype != torch.float16:
                raise RuntimeError("MixLinear_GEMM: x must be float16")
            if inputs.shape[1] != self.in_features:
                raise RuntimeError(f"MixLinear_GEMM: x has {inputs.shape[1]} columns, the layer {self.in_features} input features")
            if inputs.stride(1) != 1 and M:
                raise RuntimeError("mixq_amd.mixlib: x must be 2-D with a contiguous last dimension")
            _backend.dev_check(inputs)                       # (a CPU tensor: the product has no CPU path)
            key = self._frozen_key(cache, inputs, M)
            if self._plan_key != key and key in self._plans:
                self._plan, self._plan_key = self._plans[key], key       # (a batch size seen before, nothing else changed)
            if self._plan_key != key:
                self._plan = self._build_plan(cache, inputs, M)
                if self.config.compact_weights and self._plan is not None:
                    self.compact_weights_()                  # (e.g. after a load_state_dict into a frozen layer re-created q_weight)
                self._plan_key = self._frozen_key(cache, inputs, M)
                if self._plan is not None:
                    if len(self._plans) >= 4:
                        self._plans.pop(next(iter(self._plans)))         # (oldest entry; stale keys - another weight image, new outliers - just age out)
                    self._plans[self._plan_key] = self._plan
            plan = self._plan
            if plan is not None:
                tag = getattr(x, "_mixq_row_amax", None)     # left by the GEMM that produced x (forward_without_preconditionFusedSilu)
                if tag is not None and tag[1] is self and self._amax_dirty and tag[0] is self._amax_buf and tag[0].numel() >= M \
                        and tag[2] == x._version:            # (an in-place edit of x since the GEMM wrote it makes the maxima stale)
                    y1, cache.q_xcache, xo = plan.run(inputs, row_amax=tag[0], col_mask=self._col_mask())
                    self._amax_dirty = False                 # the quantiser cleared the buffer
                else:
                    y1, cache.q_xcache, xo = plan.run(inputs)
                if xo is not None:
                    cache.activation_outliers = xo
                cache.n_dev = plan.keep[1]
                cache.ind = self.ind
                return y1.reshape(cache.shape)
        if self.weight_only is True:
            # linear.py:178-184 (w8_a16_gemm, then `y += bias`): here the bias rides in the GEMM epilogue
            key = (id(self.q_weight), self.q_weight.data_ptr(), self.q_weight._version)
            if self._wpk is None or self._wpk_key != key:
                self._wpk, self._wpk_key = _backend.PackW8A16(self.q_weight), key
            y = _backend.W8A16Linear(inputs, self._wpk, self.scale_col, self.bias, self.out_features, self.in_features)
            return y.reshape(cache.shape)
        qmax = 2 ** (self.bit - 1) - 1

        if unfused:
            # k2 + k1 of linear.py:187-193 as one pass; the flag is the predicate of linear.py:201
            n = int(self.ind.shape[0])
            flag = None
            if self.add_outliers:
                flag = cache.flag
                flag.zero_()
            fmt = self.x_fmt(M)
            ind_buf, n_dev = self._ind_dev()
            cache.q_xcache, xo = _backend.QuantFused(inputs, ind_buf, cache.x_scale, self.bit, self._sigma_f, flag=flag, n_dev=n_dev, fmt=fmt)
            if n:
                cache.activation_outliers = xo[:, :n]
            cache.n_dev = n_dev
        elif getattr(cache, "n_dev", None) is not self._n_dev:
            cache.n_dev = None                               # whoever filled the cache did not use this layer's device count
        cache.ind = self.ind

        if self.add_outliers:
            if unfused:
                mispredicted = bool(int(cache.flag.item()))
            else:
                mispredicted = bool(cache.x_scale[0:M].max() > self.sigma / qmax)
            if mispredicted:
                ind = self.FindOutliers(inputs)
                cache.new_ind = ind
                self._append_outliers(cache, inputs, ind)
                fmt = self.x_fmt(M)
                if fmt != FMT_PLAIN:
                    cache.q_xcache = _backend.FindRowScalePacked(inputs, cache.x_scale, M, self.in_features, self.bit, fmt=fmt)
                else:
                    cache.q_xcache = _backend.FindRowScale(inputs, cache.x_scale, M, self.in_features, self.bit)
                cache.n_dev = None
            self.cnt += 1
            if self.cnt >= self.cache.stop or self.ind.shape[0] > 128:
                self.add_outliers = False

        y1 = self._gemm(cache, M, ACT_NONE)
        if self.config.compact_weights and not self.add_outliers:
            self.compact_weights_()
        return y1.reshape(cache.shape)

    @torch.no_grad()
    def forward_without_preconditionFusedSilu(self, x, cache, mul=None, amax_for=None):
        """gate_proj path (linear.py:292-376): reuse the activation quantised for up_proj, SiLU in the epilogue.
        `mul` (an extension): an fp16 [..., N] tensor multiplied in after the SiLU and the bias, i.e.
        (silu(gate(x)) + bias) * up(x) leaves the GEMM directly and the `gate_output *= up_output` pass of
        modules/fused/mlp.py:61-63 disappears.
        `amax_for` (an extension): the layer that consumes this output next (down_proj, mlp.py:66-67).  The GEMM then also leaves that
        layer's pre-pass row maxima (over the columns it does not extract), and its forward quantises in one pass instead of two."""
        inputs = x.reshape(-1, x.shape[-1])
        M = inputs.shape[0]
        if not self.forward_without_precondition_len == cache.ind.shape[0]:
            if cache.ind.shape[0]:
                ind = cache.new_ind
                weight_cache = _backend.DequantWeightCols(self.q_weight, self.scale_col, ind, self.bit)
                if self._wstore is None:
                    self._wstore = _ColStore(self.out_features, weight_cache.device,
                                             self.weight_cache if (self.weight_cache is not None and self.ind.shape[0]) else None)
                self._wstore.append(weight_cache)
                self.weight_cache = self._wstore.view() if self._wstore.n else weight_cache
                self.ind = cache.ind
                self.forward_without_precondition_len = self.ind.shape[0]
                self._silu_calls = 0
        if self.bit == 4 and not self.ind.shape[0]:
            raise RuntimeError("int4 mod should have outliers !")
        if getattr(cache, "n_dev", None) is not None:
            self._n_dev = cache.n_dev                        # the count of the layer whose activation this one shares
        target = None
        if amax_for is not None and self.bit == 8 and amax_for.in_features == self.out_features:
            wpk = self._packed_weight()
            if wpk is not None and \
                    _backend.amax_supported(M, self.out_features, self.in_features, _fmt_of(cache.q_xcache), _fmt_of(wpk)):
                target = amax_for.amax_target(M, x.device)
        extra = {} if target is None else {"row_amax": target[0], "col_mask": target[1]}
        if mul is not None:
            y1 = self._gemm(cache, M, ACT_SILU_MUL, addend=mul.reshape(-1, self.out_features), **extra)
        else:
            y1 = self._gemm(cache, M, ACT_SILU, **extra)
        self._silu_calls += 1
        if self.config.compact_weights and self._silu_calls >= self.cache.stop:
            self.compact_weights_()
        out = y1.reshape(cache.shape)
        if target is not None:
            out._mixq_row_amax = (target[0], amax_for, out._version)      # rides on the tensor handed to the consumer
        return out


_F6_NIBBLE = None


def _unpack_host_f6(packed, R):
    """Nibble-packed [R, K/2] matrix of an F6X128 image held in HOST memory (include/mixq_hip.h)."""
    global _F6_NIBBLE
    if _F6_NIBBLE is None:
        lut = torch.zeros(64, dtype=torch.uint8)
        for v, code in enumerate((0x00, 0x0c, 0x10, 0x12, 0x14, 0x15, 0x16, 0x17, 0x18)):      # E3M2 codes of 0 .. 8
            lut[code] = v & 0xF
            lut[code | 0x20] = (16 - v) & 0xF
        _F6_NIBBLE = lut
    rows16, B = packed.shape
    K = B * 4 // 3
    blocks = packed.reshape(K // 128, rows16 // 16, 1536).to(torch.int64)
    frag = torch.cat([blocks[..., :1024].reshape(K // 128, rows16 // 16, 64, 16), blocks[..., 1024:].reshape(K // 128, rows16 // 16, 64, 8)], dim=-1)
    tri = frag.reshape(K // 128, rows16 // 16, 64, 8, 3)                                     # 3 bytes = 4 codes
    v = tri[..., 0] | (tri[..., 1] << 8) | (tri[..., 2] << 16)
    codes = torch.stack([(v >> (6 * i)) & 63 for i in range(4)], dim=-1).reshape(K // 128, rows16 // 16, 4, 16, 32)   # [kb, rb, g, r, e]
    nib = _F6_NIBBLE[codes].permute(1, 3, 0, 2, 4).reshape(rows16, K)                         # [rb, r, kb, g, e] -> row, k
    return (nib[:, 0::2] | (nib[:, 1::2] << 4))[:R].contiguous()


def _unpack_host(packed, R, fmt):
    """Plain [R,KB] matrix of a packed image held in HOST memory (include/mixq_hip.h: P16X64 / F16X64 / F6X128 block layouts)."""
    if fmt == FMT_F6X128:
        return _unpack_host_f6(packed, R)
    rows16, KB = packed.shape
    blocks = packed.reshape(KB // 64, rows16 // 16, 1024)
    if fmt == FMT_F16X64:                                                # byte c*256 + r*16 + b
        t = blocks.reshape(KB // 64, rows16 // 16, 4, 16, 16).permute(1, 3, 0, 2, 4)          # [rb, r, kb, c, b]
    else:                                                                # P16X64: row r stores chunk c at position c ^ (-(r>>2) & 3)
        t = blocks.reshape(KB // 64, rows16 // 16, 16, 4, 16)
        r = torch.arange(16)
        pos = torch.arange(4).unsqueeze(0) ^ ((0 - (r >> 2)) & 3).unsqueeze(1)                # [r, c] -> physical chunk
        t = torch.gather(t, 3, pos.reshape(1, 1, 16, 4, 1).expand(KB // 64, rows16 // 16, 16, 4, 16)).permute(1, 2, 0, 3, 4)
    return t.reshape(rows16, KB)[:R].contiguous()


def _wide(t, cap):
    """View of the first `cap` columns of the storage behind a [R,n] matrix whose row stride is >= cap (the pad may hold
    anything: the kernels mask columns beyond the device-resident count)."""
    return t.as_strided((t.shape[0], cap), (t.stride(0), 1), t.storage_offset())


# north_star name for the same operator
MixQLinear = MixLinear_GEMM
