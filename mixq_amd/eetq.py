"""`EETQ`-named surface for the weight-only W8A16 layers.

The reference does `from EETQ import quant_weights, preprocess_weights, w8_a16_gemm` (modules/linear.py:9) and uses
`quant_weights(W^T.cpu(), torch.int8, False)` at :102-106 and `w8_a16_gemm(x, q_weight, scale_col)` at :178-184.  EETQ
(github.com/NetEase-FuXi/EETQ) is not part of the reference tree and not version-pinned, so parity for this row is
UNPINNED; the quantisation rule restated here is the published one it wraps (FasterTransformer's symmetric per-column
int8: scale = colabsmax / 128, q = clip(round_half_away(w / scale), -128, 127)).

Difference from EETQ, by design: in memory `q_weight` is the plain row-major int8 [K,N] matrix (EETQ's *unprocessed*
tensor): the CUTLASS-specific interleaving EETQ bakes into its checkpoints has no meaning on CDNA4, and the kernel's own
tile-major copy is built on the device on first use (`mixlib.PackW8A16`).  On DISK either layout is handled:
`preprocess_weights` / `unprocess_weights` restate EETQ's interleave and its inverse, and mixq_amd.checkpoint records
which one a checkpoint holds (`quant_config.json: "w8a16_layout"`), converting reference checkpoints on load.
`sys.modules["EETQ"] = mixq_amd.eetq` makes reference-style code run.
"""
import torch

from . import mixlib as _mixlib

_backend = _mixlib
_packed = {}          # (data_ptr, version, shape) -> (q_weight, packed weights), for callers that only hold q_weight (w8_a16_gemm).
                      # The entry keeps q_weight itself alive: a key made of an address must not outlive the allocation it names.


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


_PERM16 = (0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15)


def preprocess_weights(q_weight, *args, **kwargs):
    """EETQ's `preprocess_weights`: the plain int8 [K,N] matrix -> the CUTLASS mixed-GEMM image EETQ bakes into its checkpoints
    (what `quant_weights(..., return_unprocessed_quantized_tensor=False)` returns and the reference stores as `q_weight`,
    modules/linear.py:102-106).  Restated from the published FasterTransformer routine EETQ wraps
    (`preprocess_weights_for_mixed_gemm`, int8, sm75-sm90): (1) inside every 16 k-rows the order 0 1 8 9 2 3 10 11 4 5 12 13
    6 7 14 15 (the ldmatrix/IMMA fragment order), (2) transpose to column-major [N,K], (3) interleave column pairs per
    64-element k tile: [N/2][K/64][2][64], (4) +128 (offset binary) and swap bytes 1 and 2 of every 4.  PARITY UNPINNED: EETQ is
    not part of the reference tree and pins no version; the gfx950 kernels never read this image - it exists so reference
    weight-only checkpoints can be loaded (`unprocess_weights`) and written."""
    q = q_weight
    if q.dtype != torch.int8 or q.dim() != 2:
        raise ValueError("preprocess_weights: expected an int8 [K,N] tensor")
    K, N = q.shape
    if K % 64 or N % 2:
        raise ValueError("preprocess_weights: K must be a multiple of 64 and N even (the CUTLASS tile interleave)")
    dev = q.device
    rows = (torch.arange(K, device=dev) // 16) * 16 + torch.tensor(_PERM16, device=dev).repeat(K // 16)
    t = q[rows].t().contiguous()                                          # (1), (2): [N,K]
    t = t.reshape(N // 2, 2, K // 64, 64).permute(0, 2, 1, 3).contiguous()    # (3)
    u = (t.to(torch.int16) + 128).to(torch.uint8).reshape(-1, 4)[:, [0, 2, 1, 3]].contiguous()   # (4)
    return u.view(torch.int8).reshape(K, N)


def unprocess_weights(processed):
    """Inverse of `preprocess_weights`: a reference checkpoint's weight-only `q_weight` -> the plain int8 [K,N] matrix."""
    p = processed
    if p.dtype != torch.int8 or p.dim() != 2:
        raise ValueError("unprocess_weights: expected an int8 [K,N] tensor")
    K, N = p.shape
    if K % 64 or N % 2:
        raise ValueError("unprocess_weights: K must be a multiple of 64 and N even")
    dev = p.device
    u = p.contiguous().view(torch.uint8).reshape(-1, 4)[:, [0, 2, 1, 3]]
    t = (u.to(torch.int16) - 128).to(torch.int8).reshape(N // 2, K // 64, 2, 64).permute(0, 2, 1, 3).reshape(N, K)
    q_perm = t.t().contiguous()                                           # rows still in fragment order
    rows = (torch.arange(K, device=dev) // 16) * 16 + torch.tensor(_PERM16, device=dev).repeat(K // 16)
    q = torch.empty_like(q_perm)
    q[rows] = q_perm
    return q


def packed_weight(q_weight):
    key = (q_weight.data_ptr(), q_weight._version, tuple(q_weight.shape))
    e = _packed.get(key)
    if e is None:
        if len(_packed) > 256:
            _packed.clear()
        e = _packed[key] = (q_weight, _backend.PackW8A16(q_weight))
    return e[1]


def w8_a16_gemm(x, q_weight, scale_col, bias=None):
    """y[..., N] = x[..., K] @ (q_weight[K,N] * scale_col[N]) (+ bias): linear.py:180."""
    K, N = q_weight.shape
    y = _backend.W8A16Linear(x.reshape(-1, K), packed_weight(q_weight), scale_col.reshape(-1), bias, N, K)
    return y.reshape(x.shape[:-1] + (N,))
