"""Round 5: the kept OUTLIER MAP of a frozen layer (bits, count, per-column AND-masks) - the quantise passes take the outlier values out
of the registers that hold the row instead of gathering x[row][ind[j]] through two dependent memory round trips.  Everything here
compares the kept route with the in-kernel-mask route through the C ABI, byte for byte."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402
from mixq_amd import linear as L  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(M, K, ncols, cap, seed, scale=30.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).half()
    cols = torch.randperm(K, generator=g)[:ncols].to(torch.int32)          # unsorted, as `ind` is after an append
    x[:, cols.long()] *= scale
    ind = torch.full((cap,), K - 1, dtype=torch.int32)
    ind[:ncols] = cols
    return x, cols, ind.to(DEV)


def test_kept_outlier_map_layout():
    """Restates include/mixq_hip.h: W bit words, the count word, pad to 4 words, K 16-bit AND-masks (two per word; 0 for an outlier column)."""
    K = 200
    ind = torch.tensor([7, 199, 0, 64, 33], dtype=torch.int32, device=DEV)
    m = L.kept_outlier_map(ind, K).cpu()
    W = (K + 31) // 32
    assert m.numel() == mixlib.kept_map_words(K) == ((W + 1 + 3) // 4) * 4 + K // 2
    bits = m[:W].numpy().view("uint32")
    for c in range(K):
        assert bool((int(bits[c // 32]) >> (c % 32)) & 1) == (c in ind.tolist())
    assert int(m[W]) == 5
    pos = m[((W + 1 + 3) // 4) * 4:].numpy().view("uint16")
    for c in range(K):
        assert int(pos[c]) == (0 if c in ind.tolist() else 0xffff)


@pytest.mark.parametrize("bit,fmt,M,K,ncols,cap", [(8, 1, 512, 4096, 41, 48), (8, 1, 130, 11008, 110, 112), (8, 0, 33, 1024, 600, 608),
                                                   (4, 4, 64, 2048, 128, 128), (8, 1, 48, 28672, 287, 288), (8, 0, 5, 64, 64, 64)])
def test_quantise_kept_map_route_is_byte_identical(bit, fmt, M, K, ncols, cap):
    """Also: more outlier columns than a row has threads (600 of 1024), EVERY column an outlier (64 of 64), the widest BASELINE layer."""
    x, cols, ind = _case(M, K, ncols, cap, seed=K + ncols)
    n_dev = torch.tensor([ncols], dtype=torch.int32, device=DEV)
    kept = L.kept_outlier_map(ind[:ncols], K)
    outs = []
    for cm in (None, kept):
        xd = x.clone().to(DEV)
        sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        q, xo = mixlib.QuantFused(xd, ind, sx, bit, 6.0, flag=flag, n_dev=n_dev if cap > ncols else None, fmt=fmt, col_mask=cm)
        torch.cuda.synchronize()
        if fmt:
            q = mixlib.UnpackOperand(q, M)
        outs.append((q.clone(), sx, xo[:, :ncols].clone(), xd, flag))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert int((outs[1][3][:, cols.long().to(DEV)] != 0).sum()) == 0
    keep = torch.ones(K, dtype=torch.bool)
    keep[cols.long()] = False
    assert torch.equal(outs[1][3].cpu()[:, keep], x[:, keep])                # every other column of x is untouched


@pytest.mark.parametrize("bit,fmt,M,K", [(4, 4, 71, 2048), (4, 3, 33, 1024), (8, 2, 129, 11008), (4, 1, 5, 256)])
def test_the_slow_form_of_the_kept_route_in_every_launch_geometry(bit, fmt, M, K):
    """A kept map whose count word disagrees with the live count, through the geometries the formats select: two rows per workgroup with an odd
    row count (FP6 activations: the last workgroup holds an invalid row and still takes both barriers of the slow form), 128 / 256 / 512
    threads per row, three chunks per thread - same bytes as the route that never saw a map, and x zeroed at exactly the live columns."""
    ncols, cap = 23, 32
    x, cols, ind = _case(M, K, ncols, cap, seed=M + K)
    kept = L.kept_outlier_map(ind[:ncols], K)
    live = ncols - 4
    n_dev = torch.tensor([live], dtype=torch.int32, device=DEV)
    outs = []
    for cm in (None, kept):
        xd = x.clone().to(DEV)
        sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV)
        q, xo = mixlib.QuantFused(xd, ind, sx, bit, 6.0, n_dev=n_dev, fmt=fmt, col_mask=cm)
        torch.cuda.synchronize()
        outs.append((mixlib.UnpackOperand(q, M).clone(), sx, xo[:, :live].clone(), xd))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    zeroed = outs[1][3][:, cols[:live].long().to(DEV)]
    assert int((zeroed != 0).sum()) == 0
    assert torch.equal(outs[1][3][:, cols[live:].long().to(DEV)].cpu(), x[:, cols[live:].long()])      # the columns past the live count keep their values


@pytest.mark.parametrize("what", ["quant", "norm"])
def test_a_kept_map_built_for_another_live_count_is_ignored(what):
    """Device code may lower the live count behind the host's back: the map's count word then disagrees and the pass builds its own mask
    from ind[0 .. live) - same bytes as the route that never saw a map."""
    M, K, ncols, cap = 70, 2048, 37, 48
    x, cols, ind = _case(M, K, ncols, cap, seed=5)
    kept = L.kept_outlier_map(ind[:ncols], K)                              # describes 37 columns ...
    n_dev = torch.tensor([ncols - 5], dtype=torch.int32, device=DEV)       # ... the device says 32
    wgt = (torch.rand(K, generator=torch.Generator().manual_seed(1)) + 0.5).half().to(DEV)
    outs = []
    for cm in (None, kept):
        xd = x.clone().to(DEV)
        sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV)
        if what == "quant":
            q, xo = mixlib.QuantFused(xd, ind, sx, 8, 6.0, n_dev=n_dev, fmt=1, col_mask=cm)
            extra = xd
        else:
            extra = torch.empty_like(xd)
            q, xo = mixlib.RMSNormQuantFused(xd, wgt, extra, 1e-5, ind, sx, 8, n_dev=n_dev, fmt=1, col_mask=cm)
        torch.cuda.synchronize()
        outs.append((mixlib.UnpackOperand(q, M).clone(), sx, xo[:, :ncols - 5].clone(), extra))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("bit,fmt,M,K,ncols,cap", [(8, 1, 512, 4096, 41, 48), (8, 1, 60, 11008, 300, 304), (4, 4, 96, 1024, 128, 128)])
def test_fused_norm_kept_map_route_is_byte_identical(bit, fmt, M, K, ncols, cap):
    x, cols, ind = _case(M, K, ncols, cap, seed=K + ncols + 1)
    n_dev = torch.tensor([ncols], dtype=torch.int32, device=DEV)
    wgt = (torch.rand(K, generator=torch.Generator().manual_seed(2)) + 0.5).half().to(DEV)
    kept = L.kept_outlier_map(ind[:ncols], K)
    outs = []
    for cm in (None, kept):
        xd = x.clone().to(DEV)
        out = torch.empty_like(xd)
        sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV)
        q, xo = mixlib.RMSNormQuantFused(xd, wgt, out, 1e-5, ind, sx, bit, n_dev=n_dev if cap > ncols else None, fmt=fmt, col_mask=cm)
        torch.cuda.synchronize()
        if fmt:
            q = mixlib.UnpackOperand(q, M)
        outs.append((q.clone(), sx, xo[:, :ncols].clone(), out))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# ---- N split of a partial last round of tiles (include/mixq_hip.h: mixq_gemm_pick_split) ---------------------------------------------------
def _split_operands(M, N, K, bit, n_out, seed):
    g = torch.Generator().manual_seed(seed)
    if bit == 8:
        qx = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8)
        qw = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8)
        xp, wp = mixlib.PackOperand(qx.to(DEV), 1), mixlib.PackOperand(qw.to(DEV), 2)
    else:
        from mixq_amd.linear import pack_to_i4
        qx = pack_to_i4(torch.randint(-7, 8, (M, K), generator=g, dtype=torch.int8))
        qw = pack_to_i4(torch.randint(-8, 8, (N, K), generator=g, dtype=torch.int8))
        xp, wp = mixlib.PackOperand(qx.to(DEV), 4), mixlib.PackOperand(qw.to(DEV), 3)
    sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(DEV)
    sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(DEV)
    xo = wo = None
    if n_out:
        pad = (n_out + 15) // 16 * 16
        xo = torch.zeros((M, pad), dtype=torch.float16)
        wo = torch.zeros((N, pad), dtype=torch.float16)
        xo[:, :n_out] = torch.randn(M, n_out, generator=g).half() * 8
        wo[:, :n_out] = torch.randn(N, n_out, generator=g).half() * 0.02
        xo, wo = xo.to(DEV)[:, :n_out], wo.to(DEV)[:, :n_out]
    bias = torch.randn(N, generator=g).half().to(DEV)
    return xp, wp, sx, sw, xo, wo, bias


@pytest.mark.parametrize("M,N,K,bit,n_out,act", [(4096, 11008, 4096, 8, 41, 0), (2048, 14336, 4096, 8, 0, 1), (2048, 5120, 13824, 8, 138, 0),
                                                 (2048, 14336, 4096, 4, 128, 0), (4096, 11008, 4096, 4, 128, 1),
                                                 (4000, 11000, 4096, 8, 70, 2), (3970, 11004, 4096, 8, 3, 0)])   # ragged M, N % 8 != 0 (unstaged stores), > 64 tail columns, the multiplier form
def test_the_n_split_of_a_partial_last_round_is_bit_identical_to_one_launch(M, N, K, bit, n_out, act):
    """The automatic choice at these shapes runs two launches over disjoint column ranges (the picked tiling over its full rounds'
    columns, a cheaper tiling over the rest); mixq_gemm_set_config(-2) runs the same problem as ONE launch: identical bits, with the fp16
    outlier tail, bias and the SiLU epilogue, int8 and the FP6-coded 4-bit form."""
    lib = _capi.load()
    fmt = 2 if bit == 8 else 3
    plan = _capi.gemm_split_plan(M, N, K, bit, fmt)
    if bit == 8:
        assert plan is not None and 0 < plan[0] < N and plan[0] % 64 == 0, plan
    if plan is None:
        pytest.skip("the model prices no split for this shape in this form")
    xp, wp, sx, sw, xo, wo, bias = _split_operands(M, N, K, bit, n_out, seed=M + N + K + bit)
    addend = torch.randn((M, N), generator=torch.Generator().manual_seed(9)).half().to(DEV) if act == 2 else None
    outs = []
    try:
        for cfg in (-2, -1):
            assert lib.mixq_gemm_set_config(cfg) == 0
            y = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
            mixlib.FusedLinear(xp, wp, sx, sw, xo, wo, n_out, bias, M, N, K, bit=bit, act=act, addend=addend, out=y)
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        lib.mixq_gemm_set_config(-1)
    assert not torch.isnan(outs[1]).any()
    assert torch.equal(outs[0], outs[1])


def test_the_n_split_keeps_the_row_maximum_side_output():
    """mixq_gemm_i8_fused_amax over a split problem: the per-row maxima (atomic max of bit patterns over the unmasked columns) and Y are the
    ones of the single launch."""
    lib = _capi.load()
    M, N, K = 2048, 14336, 4096
    assert _capi.gemm_split_plan(M, N, K, 8, 2) is not None
    xp, wp, sx, sw, xo, wo, bias = _split_operands(M, N, K, 8, 0, seed=77)
    ind = torch.randperm(N, generator=torch.Generator().manual_seed(5))[:143].to(torch.int32).to(DEV)
    mask = L.kept_outlier_map(ind, N)
    outs = []
    try:
        for cfg in (-2, -1):
            assert lib.mixq_gemm_set_config(cfg) == 0
            amax = torch.zeros(M, dtype=torch.int32, device=DEV)
            y = mixlib.FusedLinear(xp, wp, sx, sw, None, None, 0, bias, M, N, K, bit=8, act=1, row_amax=amax, col_mask=mask)
            torch.cuda.synchronize()
            outs.append((y, amax))
    finally:
        lib.mixq_gemm_set_config(-1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert int(outs[1][1].max()) > 0


def test_split_plan_is_off_for_one_round_and_for_forced_tilings():
    assert _capi.gemm_split_plan(512, 11008, 4096, 8, 2) is None            # 232 tiles: one round
    assert _capi.gemm_split_plan(4096, 12288, 4096, 8, 2) is None           # 1536 tiles of 128 x 256: six full rounds
    assert _capi.gemm_split_plan(16, 11008, 4096, 8, 2) is None             # a weight stream
    n1, name = _capi.gemm_split_plan(4096, 11008, 4096, 8, 2)
    assert n1 == 10240 and name.startswith("wr")                            # five full rounds of 128 x 256 tiles, the last 768 columns apart


# ---- the metric shape against COMMITTED oracle outputs (tests/golden/g7: no oracle at run time) ----------------------------------------------
def test_metric_shape_against_the_committed_fullsize_fixture():
    """512 x 4096 -> 11008, 41 outlier columns x 20 (BASELINE.json configs[1]), rebuilt from seeds: the layer's quantised weights (scale_col
    bit-exact, row sums of q_weight exact), the quantise pass (x_scale of EVERY row, q_x / x_out of eight rows: bit-exact, kept-map and
    mask-building routes) and y of those rows (<= 2 fp16 ulp, the tolerance of tests/test_gpu_parity.py) against oracle/gen_fullsize_fixture.py's
    committed outputs."""
    import numpy as np
    from mixq_amd import MixLibCache, MixLinear_GEMM
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_fullsize_512x4096x11008.npz"))
    M, K, N = 512, 4096, 11008
    torch.manual_seed(0)
    lin = torch.nn.Linear(K, N, bias=False).half()
    cols = torch.randperm(K, generator=torch.Generator().manual_seed(1))[: round(0.01 * K)]
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(12)).half()
    x[:, cols] *= 20
    ind = torch.from_numpy(f["ind"]).to(DEV)
    assert sorted(cols.tolist()) == f["ind"].tolist()
    rows = torch.from_numpy(f["rows"].astype("int64")).to(DEV)
    cache = MixLibCache(M, device=DEV)
    layer = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=DEV)
    assert np.array_equal(layer.scale_col.cpu().numpy().reshape(-1).view(np.uint16), f["scale_col"].view(np.uint16))
    assert np.array_equal(layer.q_weight.to(torch.int32).sum(dim=1).cpu().numpy(), f["q_weight_rowsum"])
    wo = torch.zeros((N, 48), dtype=torch.float16, device=DEV)[:, :41]            # (the tail's operands are padded to 16 columns)
    mixlib.DequantWeightCols(layer.q_weight, layer.scale_col, ind, 8, out=wo)
    assert np.array_equal(wo[:, :4].cpu().numpy().view(np.uint16), f["weight_cache_head"].view(np.uint16))
    for cm in (None, L.kept_outlier_map(ind, K)):
        xd = x.clone().to(DEV)
        sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV)
        q, xo = mixlib.QuantFused(xd, ind, sx, 8, 6.0, fmt=1, col_mask=cm)
        torch.cuda.synchronize()
        assert np.array_equal(sx.cpu().numpy().reshape(-1).view(np.uint16), f["x_scale"].view(np.uint16))
        assert np.array_equal(mixlib.UnpackOperand(q, M)[rows].cpu().numpy(), f["q_x"])
        assert np.array_equal(xo[rows].cpu().numpy().view(np.uint16), f["x_out"].view(np.uint16))
        assert int((xd[:, ind.long()] != 0).sum()) == 0
        y = mixlib.FusedLinear(q, mixlib.PackOperand(layer.q_weight, 2), sx, layer.scale_col, xo, wo, int(ind.numel()), None, M, N, K)
        got, ref = y[rows].float().cpu().numpy(), f["y"].astype(np.float32)
        a = np.abs(ref)
        tol = np.maximum(2 * np.where(a > 0, 2.0 ** (np.floor(np.log2(np.maximum(a, 6e-5))) - 10), 2.0 ** -24), 1e-3)
        assert (np.abs(got - ref) <= tol).all(), float(np.abs(got - ref).max())
