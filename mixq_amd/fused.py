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

from . import mixlib as _hip_mixlib

_backend = _hip_mixlib
# gate_proj's GEMM also emits down_proj's per-row |x| maxima (include/mixq_hip.h: mixq_gemm_i8_fused_amax); False = the two-pass quantiser
FUSE_DOWN_AMAX = True


def set_backend(mod):
    global _backend
    prev, _backend = _backend, mod
    return prev


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
        fmt = nl.x_fmt(rows) if (hasattr(nl, "x_fmt") and hasattr(_backend, "PackOperand")) else 0
        n = int(nl.ind.shape[0])
        if n and hasattr(nl, "_ind_dev") and hasattr(_backend, "PackOperand"):
            ind, n_dev = nl._ind_dev()
        else:
            ind, n_dev = (nl.ind if n else None), None
        if hasattr(_backend, "PackOperand"):
            q, xo = _backend.RMSNormQuantFused(x, self.weight, output, self.variance_epsilon, ind, cache.x_scale, nl.bit,
                                               sigma=getattr(cache, "sigma_value", 6.0), fmt=fmt, n_dev=n_dev)
        else:                                                # host stand-in of the CPU tests
            q, xo = _backend.RMSNormQuantFused(x, self.weight, output, self.variance_epsilon, ind, cache.x_scale, nl.bit,
                                               sigma=getattr(cache, "sigma_value", 6.0))
        cache.q_xcache = q
        cache.activation_outliers = xo[:, :n] if n else None
        cache.n_dev = n_dev
        return output


class MixLlamaMLP(nn.Module):
    def __init__(self, gate_proj, down_proj, up_proj, MixGemmCache=None):
        super().__init__()
        self.down_proj_ = down_proj
        self.gate_proj_ = gate_proj
        self.up_proj_ = up_proj
        self.out_features = down_proj.out_features
        self.MLPCache = MixGemmCache

    @torch.no_grad()
    def forward(self, x):
        up_output = self.up_proj_(x, self.MLPCache)
        # silu(gate(x)) * up(x) in gate_proj's epilogue (the reference multiplies in a separate pass, mlp.py:61-63)
        # ... and down_proj's pre-pass row maxima leave the same epilogue (its quantiser then needs one pass over the activation)
        extra = {"amax_for": self.down_proj_} if FUSE_DOWN_AMAX else {}
        gate_output = self.gate_proj_.forward_without_preconditionFusedSilu(x, self.MLPCache, mul=up_output, **extra)
        return self.down_proj_(gate_output, None, True)
