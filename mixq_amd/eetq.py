"""`EETQ`-named surface for the weight-only W8A16 layers.

The reference does `from EETQ import quant_weights, preprocess_weights, w8_a16_gemm` (modules/linear.py:9) and uses
`quant_weights(W^T.cpu(), torch.int8, False)` at :102-106 and `w8_a16_gemm(x, q_weight, scale_col)` at :178-184.  EETQ
(github.com/NetEase-FuXi/EETQ) is not part of the reference tree and not version-pinned, so parity for this row is
UNPINNED; the quantisation rule restated here is the published one it wraps (FasterTransformer's symmetric per-column
int8: scale = colabsmax / 128, q = clip(round_half_away(w / scale), -128, 127)).

Difference from EETQ, by design: `q_weight` stays the plain row-major int8 [K,N] matrix (EETQ's *unprocessed* tensor);
the CUTLASS-specific interleaving EETQ bakes into the checkpoint has no meaning on CDNA4.  The kernel's own tile-major
copy is built on the device on first use (`mixlib.PackW8A16`).  `sys.modules["EETQ"] = mixq_amd.eetq` makes
reference-style code run.
"""
import torch

from . import mixlib as _mixlib

_backend = _mixlib
_packed = {}          # (data_ptr, version, shape) -> packed weights, for callers that only hold q_weight (w8_a16_gemm)


def set_backend(module):
    """Tests on machines without a GPU install a CPU-oracle backend here (test infrastructure)."""
    global _backend
    _backend = module
    _packed.clear()


def quant_weights(weight, dtype=torch.int8, return_unprocessed_quantized_tensor=False):
    """weight: [K,N] (W^T) on the CPU or GPU -> (q int8 [K,N], scales [N] in weight's dtype)."""
    if dtype != torch.int8:
        raise NotImplementedError("quant_weights: only torch.int8 is on this path (linear.py:104)")
    if weight.dim() != 2:
        raise ValueError("quant_weights: expected a 2-D [K,N] weight")
    w = weight.float()
    scale = w.abs().amax(dim=0) * (1.0 / 128.0)                     # float, per output column
    q = torch.where(scale > 0, w / scale, torch.zeros_like(w))
    q = torch.sign(q) * torch.floor(q.abs() + 0.5)                  # C round(): half away from zero
    q = q.clamp_(-128, 127).to(torch.int8)
    return q, scale.to(weight.dtype)


def preprocess_weights(q_weight, *args, **kwargs):
    """EETQ permutes for its CUTLASS kernels here; the gfx950 kernel re-tiles on first use instead."""
    return q_weight


def packed_weight(q_weight):
    key = (q_weight.data_ptr(), q_weight._version, tuple(q_weight.shape))
    p = _packed.get(key)
    if p is None:
        if len(_packed) > 256:
            _packed.clear()
        p = _backend.PackW8A16(q_weight)
        _packed[key] = p
    return p


def w8_a16_gemm(x, q_weight, scale_col, bias=None):
    """y[..., N] = x[..., K] @ (q_weight[K,N] * scale_col[N]) (+ bias): linear.py:180."""
    K, N = q_weight.shape
    y = _backend.W8A16Linear(x.reshape(-1, K), packed_weight(q_weight), scale_col.reshape(-1), bias, N, K)
    return y.reshape(x.shape[:-1] + (N,))
