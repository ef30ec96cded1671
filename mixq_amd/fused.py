"""The two modules either side of the path, mirrored from the reference so its Llama glue
(`mixquant/models/llama.py:10-22,173-178`) can wire them unchanged:

  FasterTransformerRMSNorm  (mixquant/modules/fused/norm.py:6-39): RMSNorm whose `next_layer` (the fused QKV or the up
      projection) gets its activation extracted + quantised in the same pass, so that linear runs with unfused=False.
  MixLlamaMLP               (mixquant/modules/fused/mlp.py:37-70): up_proj, the SiLU-fused gate_proj sharing up_proj's
      quantised activation, elementwise product, down_proj(unfused=True).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import linear as _L
from . import mixlib as _hip_mixlib

_backend = _hip_mixlib
# (the switches of rounds 1-4 - FUSE_DOWN_AMAX, JOINT_GATE_UP, NORM_KEPT_MASK - are fields of the model's MixqConfig now: config.py)


def interleave_pair_rows(up: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """Rows of up_proj's and gate_proj's per-channel tensors ([N, ...] each, N even) in the order MIXQ_ACT_SILU_PAIR reads them
    (include/mixq_hip.h): groups of four - up[2g], up[2g+1], gate[2g], gate[2g+1] - so that the four consecutive channels a lane holds
    of an accumulator block are two complete (up, gate) pairs."""
    if up.shape != gate.shape or up.shape[0] % 2:
        raise ValueError("interleave_pair_rows: two tensors of one shape with an even number of rows")
    N, rest = up.shape[0], tuple(up.shape[1:])
    return torch.stack([up.reshape(N // 2, 2, *rest), gate.reshape(N // 2, 2, *rest)], dim=1).reshape(2 * N, *rest)


def split_pair_rows(joint: torch.Tensor):
    """Inverse of interleave_pair_rows: (up, gate)."""
    N2, rest = joint.shape[0], tuple(joint.shape[1:])
    v = joint.reshape(N2 // 4, 2, 2, *rest)
    return v[:, 0].reshape(N2 // 2, *rest), v[:, 1].reshape(N2 // 2, *rest)


class FasterTransformerRMSNorm(nn.Module):
    def __init__(self, weight, eps=1e-6, cache=None):
        super().__init__()
        self.weight = weight.to(torch.float16)
        self.variance_epsilon = eps
        self.cache = cache
        self.next_layer = None

    @torch.no_grad()
    def forward(self, x):
        output = torch.empty_like(x)
        nl = self.next_layer
        if nl is None:
            _backend.layernorm_forward_cuda(x, self.weight, output, self.variance_epsilon)
            return output
        if nl.bit not in (4, 8):
            raise NotImplementedError
        cache = self.cache
        # the layout and the outlier bookkeeping (capacity-padded `ind`, device-resident count) are the NEXT layer's
        rows = x.numel() // x.shape[-1]                      # (4-bit layers take small batches in another layout than large ones)
        # (`next_layer` may be another implementation of the operator - anything with `.bit` and `.ind`: it then gets plain operands)
        ours = isinstance(nl, _L.MixLinear_GEMM)
        fmt = nl.x_fmt(rows) if ours else 0
        n = int(nl.ind.shape[0])
        ind, n_dev = nl._ind_dev() if (n and ours) else ((nl.ind if n else None), None)
        # (a next layer whose prediction is frozen hands over its kept outlier map: no mask build in front of the row maximum)
        cm = nl._col_mask() if (ours and n and nl.config.norm_kept_map and not nl.add_outliers) else None
        q, xo = _backend.RMSNormQuantFused(x, self.weight, output, self.variance_epsilon, ind, cache.x_scale, nl.bit,
                                           sigma=getattr(cache, "sigma_value", 6.0), fmt=fmt, n_dev=n_dev, col_mask=cm)
        cache.q_xcache = q
        cache.activation_outliers = xo[:, :n] if n else None
        cache.n_dev = n_dev
        return output


class _JointRows:
    """What a layer's `_Derived.joint` holds once its weights live in the MLP block's joint image: how to get its own rows back."""

    __slots__ = ("owner", "which")

    def __init__(self, owner, which):
        self.owner, self.which = owner, which

    def plain_rows(self):
        j = self.owner._joint
        if j is None:
            raise RuntimeError("MixLlamaMLP: the joint gate / up image this layer's weights live in is gone")
        w = j["wpk"]
        plain = _backend.UnpackOperand(w, j["rows"]) if w.is_cuda else _L._unpack_host(w, j["rows"], _L._fmt_of(w))
        return split_pair_rows(plain)[self.which].contiguous()


class MixLlamaMLP(nn.Module):
    def __init__(self, gate_proj, down_proj, up_proj, MixGemmCache=None):
        super().__init__()
        self.down_proj_ = down_proj
        self.gate_proj_ = gate_proj
        self.up_proj_ = up_proj
        self.out_features = down_proj.out_features
        self.MLPCache = MixGemmCache
        # the model's switches: the cache's MixqConfig when there is one, else up_proj's (config.py)
        object.__setattr__(self, "config", getattr(MixGemmCache, "config", None) or getattr(up_proj, "config", None) or _L.MixqConfig())
        # gate_proj and up_proj as ONE launch (MIXQ_ACT_SILU_PAIR): the interleaved operands, built at the first forward after both
        # layers' outlier predictions froze.  A plain attribute, not a buffer: the state_dict stays the reference's (two q_weight tensors)
        object.__setattr__(self, "_joint", None)
        object.__setattr__(self, "_two_launch_captured", False)   # a hipGraph out there replays the two layers' OWN images

    # ---- gate_proj + up_proj in one launch -------------------------------------------------------------------------------
    def _joint_key(self):
        up, gate = self.up_proj_, self.gate_proj_

        def ident(t):
            return None if t is None else (id(t), t._version, tuple(t.shape))
        return tuple((ident(l._buffers.get("q_weight")), ident(l.scale_col), ident(l.bias), ident(l.weight_cache), ident(l.ind))
                     for l in (up, gate)) + (str(up.scale_col.device),) + self.config.key()

    def _joint_applies(self, cache):
        up, gate = self.up_proj_, self.gate_proj_
        if not self.config.joint_gate_up or not _backend.PAIR_LAUNCH:
            return False
        if up.bit != gate.bit or up.weight_only or gate.weight_only or up.add_outliers:
            return False
        # the kernels with the paired epilogue: int8 on fragment-order weights, int4 as FP6 codes (the default image of either width)
        if (self.config.pack_fmt != _L.FMT_F16X64) if up.bit == 8 else (self.config.pack_fmt4 != _L.FMT_F6X128):
            return False
        if up.in_features % (64 if up.bit == 8 else 128) or up.out_features % 8 or \
                (up.in_features, up.out_features) != (gate.in_features, gate.out_features):
            return False
        n = int(up.ind.shape[0])
        if up.bit == 4 and n == 0:
            return False                                     # (the two-launch route raises "int4 mod should have outliers", linear.py:255)
        if n and (gate.forward_without_precondition_len != n or gate.weight_cache is None or gate.weight_cache.shape[1] != n
                  or up.weight_cache is None or up.weight_cache.shape[1] != n):
            return False                                     # (gate_proj has not taken over the latest outlier columns yet: its own route does that)
        if int(gate.ind.shape[0]) != n:
            return False
        if _L._fmt_of(cache.q_xcache) != (_L.FMT_P16X64 if up.bit == 8 else _L.FMT_R6X128):
            return False
        j = self._joint
        key = self._joint_key()
        object.__setattr__(self, "_key_now", key)            # (handed to _joint_operands: one identity sweep per forward)
        if (j is None or j["key"] != key) and _L._capturing():
            # the joint image is never BUILT under capture (its packing kernels would be replayed with the graph, and the layers' own
            # images - which that graph is about to address through the two-launch route - would be freed by the build)
            object.__setattr__(self, "_two_launch_captured", True)
            return False
        if (j is None or j["key"] != key) and n and gate.ind is not up.ind and not torch.equal(gate.ind, up.ind):
            # the joint launch multiplies up_proj's activation outliers (extracted in `up.ind` order) with BOTH layers' weight_cache columns:
            # a gate_proj holding another column set or order (a per-layer load_state_dict of `ind` / `weight_cache`) keeps the two launches
            # (checked when the image is (re)built, not per forward: it is a host sync; `ind`'s identity and version are part of the key)
            return False
        return True

    def _joint_operands(self):
        """The interleaved image + per-channel operands (interleave_pair_rows), rebuilt when anything they were made from changed.  The
        two layers then give up their own packed images: the block keeps ONE copy of these weights."""
        key = self.__dict__.pop("_key_now", None) or self._joint_key()
        j = self._joint
        if j is not None and j["key"] == key:
            return j
        up, gate = self.up_proj_, self.gate_proj_
        N, n = up.out_features, int(up.ind.shape[0])
        wpk = _backend.PackOperand(interleave_pair_rows(up.q_weight, gate.q_weight), _L.FMT_F16X64 if up.bit == 8 else _L.FMT_F6X128)
        scale = interleave_pair_rows(up.scale_col.reshape(-1), gate.scale_col.reshape(-1)).reshape(1, -1)
        bias = None
        if up.bias is not None or gate.bias is not None:
            zero = torch.zeros((N,), dtype=torch.float16, device=scale.device)
            bias = interleave_pair_rows(zero if up.bias is None else up.bias, zero if gate.bias is None else gate.bias)
        wo = None
        if n:
            wo = torch.zeros((2 * N, _L._pad16(n)), dtype=torch.float16, device=scale.device)
            wo[:, :n] = interleave_pair_rows(up.weight_cache, gate.weight_cache)
            wo = wo[:, :n]
        old = self._joint
        j = {"wpk": wpk, "rows": 2 * N, "scale": scale, "bias": bias, "wo": wo,
             # a superseded joint image may still be addressed by a graph captured on this route: kept alive (rebuilds are rare: new weights
             # loaded, new outlier columns, a device move)
             # - only when such a graph exists; otherwise every load_state_dict / outlier change of a frozen block would pin one more full
             # image (90 MB per Llama-2-7b block)
             "retired": ([] if old is None else old["retired"] + ([old["wpk"], old["wo"], old["scale"], old["bias"]] if old.get("captured") else []))}
        object.__setattr__(self, "_joint", j)
        for which, l in enumerate((up, gate)):
            d = l._d
            if d.wpk is not None and (self._two_launch_captured or any(getattr(pl, "captured", False) for pl in d.plans.values())):
                d.retired.append(d.wpk)                      # (a captured graph still addresses the layer's own image: kept alive, else freed here)
            d.invalidate()                                   # (kept argument blocks carry the old image's address)
            d.wpk = d.wpk_key = None
            d.joint = _JointRows(self, which)
            if self.config.compact_weights:
                l._buffers["q_weight"] = None
        j["key"] = self._joint_key()
        return j

    def _apply(self, fn, *args, **kwargs):
        j = self._joint
        if j is not None and j.get("captured") and not getattr(self, "allow_move_after_capture", False) and j["wpk"].is_cuda:
            here = j["wpk"].device
            if fn(torch.empty(0, device=here)).device != here:
                # (as MixLinear_GEMM._apply: a hipGraph captured through the joint route replays the raw address of the joint image)
                raise RuntimeError("MixLlamaMLP: this block ran under hipGraph capture; moving it to another device would leave that graph with "
                                   "dangling addresses.  Drop the graph first, then set block.allow_move_after_capture = True (re-capture after "
                                   "the move).")
        out = super()._apply(fn, *args, **kwargs)
        j = self._joint
        if j is not None:                                    # the joint image is the only copy of gate_proj's / up_proj's weights: it follows the module
            w = fn(j["wpk"])
            if w is not j["wpk"]:
                _backend.set_fmt(w, _L._fmt_of(j["wpk"]))
                j["wpk"], j["key"], j["retired"] = w, None, []   # (the per-channel operands are re-made from the moved layers)
            # (an _apply that moves nothing - model.half() on an fp16 block - leaves the image, its key and the buffers a captured graph
            # addresses alone; changed per-channel tensors show up in the key on the next forward)
        return out

    def _forward_joint(self, x, cache):
        up, gate, down = self.up_proj_, self.gate_proj_, self.down_proj_
        inputs = x.reshape(-1, x.shape[-1])
        M, N, K = inputs.shape[0], up.out_features, up.in_features
        cache.shape = x.shape[:-1] + (N,)                    # (what up_proj's forward leaves: linear.py:169-172)
        if getattr(cache, "n_dev", None) is not up._n_dev:
            cache.n_dev = None
        cache.ind = up.ind
        j = self._joint_operands()
        for l in (up, gate):
            # a direct call of up_proj / gate_proj since the last joint forward re-packed a private image from the joint one: released
            # here (the block keeps ONE copy of these weights) unless a captured graph addresses it
            d = l._d
            if d.joint is not None and (d.wpk is not None or d.wpk_small is not None) and not _L._capturing() \
                    and not any(getattr(pl, "captured", False) for pl in d.plans.values()):
                d.invalidate()
                d.wpk = d.wpk_key = None
                d.wpk_small = d.wpk_small_key = None
        if not j.get("captured") and _L._capturing():
            j["captured"] = True                             # a graph out there replays the joint image's address (see _apply)
        n = int(up.ind.shape[0])
        xo = wo = n_dev = None
        n_cap = 0
        if n:
            xo = cache.activation_outliers
            if xo is None or xo.shape[1] != n:
                raise RuntimeError("MixLlamaMLP: outlier operands do not match `ind`")
            xo, wo = _L._gemm_ready(xo), j["wo"]
            cap = min(_L._pad16(n), xo.stride(0), wo.stride(0))
            if cache.n_dev is not None and cap > n:
                n_dev, n_cap = up._n_dev, cap
            else:
                n_cap = n
            xo, wo = _L._wide(xo, n_cap), _L._wide(wo, n_cap)
        extra = {}
        target = None
        if self.config.fuse_down_amax and up.bit == 8 and down.bit == 8 and down.in_features == N and \
                _backend.amax_supported(M, 2 * N, K, _L.FMT_P16X64, _L.FMT_F16X64):
            target = down.amax_target(M, x.device)
            if target is not None:
                extra = {"row_amax": target[0], "col_mask": target[1]}
        y = _backend.FusedLinear(cache.q_xcache, j["wpk"], cache.x_scale, j["scale"], xo, wo, n_cap, j["bias"], M, 2 * N, K, bit=up.bit,
                                 act=_hip_mixlib.ACT_SILU_PAIR, n_out_dev=n_dev, **extra)
        out = y.reshape(cache.shape)
        if target is not None:
            out._mixq_row_amax = (target[0], down, out._version)
        return out

    @torch.no_grad()
    def forward(self, x):
        rows = x.numel() // x.shape[-1]
        if not self.up_proj_._small_batch_image(rows) and self._joint_applies(self.MLPCache):
            # silu(gate(x)) * up(x) from ONE launch over the two layers' interleaved rows (mlp.py:57-63), bit-identical to the route below
            return self.down_proj_(self._forward_joint(x, self.MLPCache), None, True)
        up_output = self.up_proj_(x, self.MLPCache)
        # silu(gate(x)) * up(x) in gate_proj's epilogue (the reference multiplies in a separate pass, mlp.py:61-63)
        # ... and down_proj's pre-pass row maxima leave the same epilogue (its quantiser then needs one pass over the activation)
        extra = {"amax_for": self.down_proj_} if self.config.fuse_down_amax else {}
        gate_output = self.gate_proj_.forward_without_preconditionFusedSilu(x, self.MLPCache, mul=up_output, **extra)
        return self.down_proj_(gate_output, None, True)


class MixFalconMLP(nn.Module):
    """mixquant/modules/fused/mlp.py:8-32 (models/falcon.py:53): dense_h_to_4h -> exact GELU -> dense_4h_to_h with the unfused pre-pass.  The
    two Linears are this package's operator (quantise + GEMM kernels); the GELU between them is the reference's own torch module."""

    def __init__(self, dense_h_to_4h, dense_4h_to_h, MixGemmCache=None):
        super().__init__()
        self.dense_h_to_4h = dense_h_to_4h
        self.dense_4h_to_h = dense_4h_to_h
        self.act = nn.GELU()
        self.MixGemmCache = MixGemmCache

    def forward(self, x):
        x = self.act(self.dense_h_to_4h(x, self.MixGemmCache))
        return self.dense_4h_to_h(x, self.MixGemmCache, True)


class MixGPTJMLP(nn.Module):
    """mixquant/modules/fused/mlp.py:75-93 (models/gptj.py:56): fc_in -> config.activation_function -> fc_out -> dropout, the layers called
    with an MLPCache (Cache.py:42-48).  Deviation: the reference always builds a bare MLPCache and ignores its MixGemmCache argument - enough
    for weight-only layers (fc_out under the GPT-J policy, utils/module.py:4-12), but a mixed-precision fc_in reads `sigma`, `zeros`, ... that an
    MLPCache does not have; here the model's MixLibCache is used when one is given."""

    def __init__(self, module, config, MixGemmCache=None):
        super().__init__()
        from transformers.activations import ACT2FN
        from .cache import MLPCache
        self.fc_in = module.fc_in
        self.fc_out = module.fc_out
        self.act = ACT2FN[config.activation_function]
        self.dropout = nn.Dropout(config.resid_pdrop)
        dev = next((b.device for b in self.fc_in.buffers()), torch.device("cpu"))
        self.MLPCache = MixGemmCache if MixGemmCache is not None else MLPCache(device=dev)

    def forward(self, hidden_states):
        hidden_states = self.fc_in(hidden_states, self.MLPCache)
        hidden_states = self.act(hidden_states)
        hidden_states = self.fc_out(hidden_states, self.MLPCache)
        return self.dropout(hidden_states)
