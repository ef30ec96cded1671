"""gate_proj + up_proj of an MLP block as ONE launch (MIXQ_ACT_SILU_PAIR, include/mixq_hip.h; mixquant/modules/fused/mlp.py:57-63): the
interleaved weight image against the oracle and, bit for bit, against the two-launch route it replaces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mixq_amd import _capi, mixlib  # noqa: E402
from mixq_amd.fused import interleave_pair_rows, split_pair_rows  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_parity import make_x, t  # noqa: E402
from test_gpu_round3 import n, ulp_tol  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    assert "gfx950" in _capi.device_info()
    _capi.load().mixq_gemm_set_config(-1)
    yield
    _capi.load().mixq_gemm_set_config(-1)


def _case(M, N, K, n_out, bias, seed, bit=8):
    rng = np.random.default_rng(seed)
    ind = np.sort(rng.choice(K, n_out, replace=False)).astype(np.int32) if n_out else np.zeros(0, np.int32)
    x = make_x(M, K, seed=seed + 1, outlier_cols=ind)
    c = dict(M=M, N=N, K=K, ind=ind, bit=bit)
    for nm in ("up", "gate"):
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
        if bit == 8:
            qw, sw = O.quant_weight_w8(w)
            wo = O.dequant_weight_cols(qw, sw, ind, 8) if n_out else None
        else:
            qw, sw, wo = O.quant_weight_w4(w, ind)                  # fp columns keep their exact fp16 weights (linear.py:129)
            wo = wo if n_out else None
        c[nm] = dict(qw=qw, sw=sw, wo=wo, bias=rng.standard_normal(N).astype(np.float16) if bias else None)
    xz = x.copy()
    c["xo"] = O.extract_outliers_zero(xz, ind) if n_out else None
    c["qx"], c["sx"] = O.find_row_scale(xz, bit)
    return c


def _fmts(c):
    return (1, 2) if c["bit"] == 8 else (_capi.FMT_R6X128, _capi.FMT_F6X128)       # (activations, weights)


def _operands(c):
    M, n_out = c["M"], int(c["ind"].size)
    pad = (n_out + 15) // 16 * 16
    xo = None
    if n_out:
        xo = torch.full((M, pad), float("nan"), dtype=torch.float16, device=DEV)   # the pad is poison
        xo[:, :n_out] = t(c["xo"])
        xo = xo[:, :n_out]
    sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV); sx[:, 0] = t(c["sx"])
    qx = mixlib.PackOperand(t(c["qx"]), _fmts(c)[0])

    def wo_of(w):
        if not n_out:
            return None
        b = torch.full((w.shape[0], pad), float("nan"), dtype=torch.float16, device=DEV)
        b[:, :n_out] = w
        return b[:, :n_out]
    return qx, sx, xo, wo_of, n_out


def _two_launches(c, row_amax=None, col_mask=None):
    M, N, K = c["M"], c["N"], c["K"]
    qx, sx, xo, wo_of, n_out = _operands(c)
    u, g = c["up"], c["gate"]
    b = lambda d: None if d["bias"] is None else t(d["bias"])
    wf = _fmts(c)[1]
    up = mixlib.FusedLinear(qx, mixlib.PackOperand(t(u["qw"]), wf), sx, t(u["sw"]), xo, wo_of(t(u["wo"])) if n_out else None, n_out, b(u), M, N, K,
                            bit=c["bit"])
    extra = {} if row_amax is None else {"row_amax": row_amax, "col_mask": col_mask}
    return mixlib.FusedLinear(qx, mixlib.PackOperand(t(g["qw"]), wf), sx, t(g["sw"]), xo, wo_of(t(g["wo"])) if n_out else None, n_out, b(g), M, N, K,
                              bit=c["bit"], act=_capi.ACT_SILU_MUL, addend=up, **extra)


def _one_launch(c, row_amax=None, col_mask=None):
    M, N, K = c["M"], c["N"], c["K"]
    qx, sx, xo, wo_of, n_out = _operands(c)
    u, g = c["up"], c["gate"]
    qw = mixlib.PackOperand(interleave_pair_rows(t(u["qw"]), t(g["qw"])), _fmts(c)[1])
    sw = interleave_pair_rows(t(u["sw"]).reshape(-1), t(g["sw"]).reshape(-1)).reshape(1, -1)
    wo = wo_of(interleave_pair_rows(t(u["wo"]), t(g["wo"]))) if n_out else None
    bias = None if u["bias"] is None else interleave_pair_rows(t(u["bias"]), t(g["bias"]))
    extra = {} if row_amax is None else {"row_amax": row_amax, "col_mask": col_mask}
    return mixlib.FusedLinear(qx, qw, sx, sw, xo, wo, n_out, bias, M, 2 * N, K, bit=c["bit"], act=_capi.ACT_SILU_PAIR, **extra)


def test_interleave_is_a_bijection_in_groups_of_four():
    up = torch.arange(16).reshape(8, 2)
    gate = 100 + torch.arange(16).reshape(8, 2)
    j = interleave_pair_rows(up, gate)
    assert j[:, 0].tolist() == [0, 2, 100, 102, 4, 6, 104, 106, 8, 10, 108, 110, 12, 14, 112, 114]
    u2, g2 = split_pair_rows(j)
    assert torch.equal(u2, up) and torch.equal(g2, gate)


@pytest.mark.parametrize("M,N,K,n_out,bias", [(512, 11008, 4096, 41, False), (200, 584, 512, 70, True), (130, 104, 256, 33, True),
                                              (96, 96, 128, 0, False), (33, 1000, 1024, 129, False), (512, 4096, 1024, 20, False)])
def test_one_launch_for_gate_and_up_against_the_oracle_and_the_two_launch_route(M, N, K, n_out, bias):
    c = _case(M, N, K, n_out, bias, seed=M + N + n_out)
    u, g = c["up"], c["gate"]
    up_ref = O.linear_fused(c["qx"], u["qw"], c["sx"], u["sw"], xo=c["xo"], wo=u["wo"], addend=None, bias=u["bias"], act=0, bit=8)
    ref = O.linear_fused(c["qx"], g["qw"], c["sx"], g["sw"], xo=c["xo"], wo=g["wo"], addend=up_ref, bias=g["bias"], act=2, bit=8).astype(np.float32)
    y1 = _one_launch(c)
    assert tuple(y1.shape) == (M, N)
    y = n(y1).astype(np.float32)
    assert np.isfinite(y).all()
    assert (np.abs(y - ref) <= ulp_tol(ref)).all(), float(np.abs(y - ref).max())
    if M * N <= 2e6:
        # the oracle's paired step on the INTERLEAVED operands (its own, element-by-element definition of the row order): the same bits
        # as its two steps, and the image the product builds is that definition
        il = O.pair_rows_interleave
        ref2 = O.linear_fused_pair(c["qx"], il(u["qw"], g["qw"]), c["sx"], il(np.asarray(u["sw"]).reshape(-1), np.asarray(g["sw"]).reshape(-1)),
                                   xo=c["xo"], wo2=None if u["wo"] is None else il(u["wo"], g["wo"]),
                                   bias2=None if u["bias"] is None else il(u["bias"], g["bias"]))
        assert np.array_equal(ref2.astype(np.float32), ref)
        assert np.array_equal(n(interleave_pair_rows(t(u["qw"]), t(g["qw"]))), il(u["qw"], g["qw"]))
    y2 = _two_launches(c)
    assert torch.equal(y1, y2), int((y1 != y2).sum())
    if M * N > 2e6:
        return
    lib, names = _capi.load(), _capi.gemm_config_names()                # every tiling that has the paired epilogue, forced
    forms = ("wr128x192_s16_d4_l2", "wr128x256_s16_d3_l2", "wr64x192_s16_d4_l2", "wr64x256_s16_d4_l2", "wr32x64_s8_d6_l1")
    try:
        for nm in forms:
            assert lib.mixq_gemm_set_config(names.index(nm)) == 0
            y3 = _one_launch(c)
            assert torch.equal(y3, y2), (nm, int((y3 != y2).sum()))
        assert lib.mixq_gemm_set_config(names.index("wr128x128_s16_d4_l2")) == 0      # a tiling without it: refused
        with pytest.raises(_capi.MixqError):
            _one_launch(c)
    finally:
        lib.mixq_gemm_set_config(-1)


def test_row_maxima_for_down_proj_leave_the_joint_launch_as_they_leave_gate_projs():
    M, N, K, n_out = 300, 1024, 512, 41
    c = _case(M, N, K, n_out, False, seed=7)
    mask = torch.zeros((N + 31) // 32, dtype=torch.int32, device=DEV)
    for col in (0, 5, 33, 64, 1001, 1023):                               # down_proj's own outlier columns: out of its row maxima
        mask[col // 32] |= (1 << (col % 32)) if col % 32 != 31 else -(1 << 31)
    a1 = torch.zeros(M, dtype=torch.int32, device=DEV)
    a2 = torch.zeros(M, dtype=torch.int32, device=DEV)
    y1 = _one_launch(c, a1, mask)
    y2 = _two_launches(c, a2, mask)
    assert torch.equal(y1, y2)
    assert torch.equal(a1, a2)
    keep = torch.ones(N, dtype=torch.bool, device=DEV)
    keep[[0, 5, 33, 64, 1001, 1023]] = False
    want = y1[:, keep].abs().max(dim=1).values.view(torch.int16).to(torch.int32)
    assert torch.equal(a1, want)


def test_operands_the_joint_form_does_not_serve_are_refused_not_miscomputed():
    c = _case(64, 96, 128, 0, False, seed=3)
    qx, sx, _, _, _ = _operands(c)
    u, g = c["up"], c["gate"]
    j = interleave_pair_rows(t(u["qw"]), t(g["qw"]))
    sw = interleave_pair_rows(t(u["sw"]).reshape(-1), t(g["sw"]).reshape(-1)).reshape(1, -1)
    with pytest.raises(_capi.MixqError):                                 # P16x64 weights: the LDS-staged kernels have no paired epilogue
        mixlib.FusedLinear(qx, mixlib.PackOperand(j, 1), sx, sw, None, None, 0, None, 64, 192, 128, act=_capi.ACT_SILU_PAIR)
    with pytest.raises(RuntimeError):
        mixlib.FusedLinear(qx, mixlib.PackOperand(j, 2), sx, sw, None, None, 0, None, 64, 192, 128, act=_capi.ACT_SILU_PAIR,
                           addend=torch.zeros((64, 96), dtype=torch.float16, device=DEV))


# ---------------------------------------------------------------------------------------------------------------
# the MLP block (mixquant/modules/fused/mlp.py:37-70) on the joint route
# ---------------------------------------------------------------------------------------------------------------
def _block(M, H, F, bias, seed=0):
    from mixq_amd import FasterTransformerRMSNorm, MixLibCache, MixLinear_GEMM, MixLlamaMLP
    torch.manual_seed(seed)
    cache = MixLibCache(M, device=DEV)
    mk = lambda k, nn_: MixLinear_GEMM.from_linear(torch.nn.Linear(k, nn_, bias=bias).half(), 8, cache=cache, dev=DEV)
    gate, up, down = mk(H, F), mk(H, F), mk(F, H)
    inner = MixLlamaMLP(gate, down, up, cache)
    norm = FasterTransformerRMSNorm((torch.rand(H) + 0.5).half().to(DEV), 1e-5, cache)
    norm.next_layer = up
    g = torch.Generator().manual_seed(seed + 1)
    cols = torch.randperm(H, generator=g)[:5]
    xs = []
    for _ in range(5):
        x = torch.randn(M, H, generator=g).half()
        x[:, cols] *= 20
        xs.append(x)
    return inner, norm, cache, xs


@pytest.mark.parametrize("M,bias", [(96, True), (512, False), (20, False)])
def test_mlp_block_on_the_joint_route_is_bit_identical_and_keeps_one_weight_image(M, bias):
    from mixq_amd import fused
    H, F = 512, 1536
    inner, norm, cache, xs = _block(M, H, F, bias)
    up, gate, down = inner.up_proj_, inner.gate_proj_, inner.down_proj_
    mlp = lambda x: inner(norm(x))
    prev = inner.config.joint_gate_up
    try:
        inner.config.joint_gate_up = False
        for x in xs[:3]:
            mlp(x.clone().to(DEV))                                    # freeze every layer's outlier search
        assert not up.add_outliers and not down.add_outliers
        q_up, q_gate = up.q_weight.clone(), gate.q_weight.clone()     # (re-created from the layers' own images)
        y_ref = [mlp(x.clone().to(DEV)) for x in xs[3:]]
        sx_ref = cache.x_scale[:M].clone()
        assert inner._joint is None and up._wpk is not None
        inner.config.joint_gate_up = True
        y = [mlp(x.clone().to(DEV)) for x in xs[3:]]
        assert inner._joint is not None, "the joint route did not run"
        assert all(torch.equal(a, b) for a, b in zip(y, y_ref)) and torch.equal(cache.x_scale[:M], sx_ref)
        # ONE copy of gate_proj's / up_proj's weights: the interleaved image; the state_dict is still the reference's
        assert up._wpk is None and gate._wpk is None and up._buffers["q_weight"] is None and gate._buffers["q_weight"] is None
        sd = inner.state_dict()
        assert torch.equal(sd["up_proj_.q_weight"], q_up) and torch.equal(sd["gate_proj_.q_weight"], q_gate)
        # under hipGraph replay
        side = torch.cuda.Stream()
        xg = xs[4].clone().to(DEV)
        keep = xg.clone()
        with torch.cuda.stream(side):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                yg = mlp(xg)
            for _ in range(3):
                xg.copy_(keep)
                gr.replay()
                torch.cuda.synchronize()
                assert torch.equal(yg, y_ref[1])
        # a layer used on its own again gets its own image back (from the joint one) and computes what it always did
        inner.config.joint_gate_up = False
        y2 = mlp(xs[3].clone().to(DEV))
        assert torch.equal(y2, y_ref[0]) and up._wpk is not None
    finally:
        inner.config.joint_gate_up = prev


def test_new_weights_loaded_into_a_block_on_the_joint_route_reach_the_joint_image():
    M, H, F = 64, 256, 512
    a, norm_a, cache_a, xs = _block(M, H, F, False, seed=0)
    b, norm_b, cache_b, _ = _block(M, H, F, False, seed=5)            # other weights
    for blk, nrm in ((a, norm_a), (b, norm_b)):
        for x in xs[:3]:
            blk(nrm(x.clone().to(DEV)))
    assert a._joint is not None and b._joint is not None
    norm_b.weight.copy_(norm_a.weight)
    y_a = a(norm_a(xs[3].clone().to(DEV)))
    y_b0 = b(norm_b(xs[3].clone().to(DEV)))
    assert not torch.equal(y_a, y_b0)
    sd = a.state_dict()
    missing, unexpected = b.load_state_dict(sd, strict=False)
    assert not unexpected
    for la, lb in ((a.up_proj_, b.up_proj_), (a.gate_proj_, b.gate_proj_), (a.down_proj_, b.down_proj_)):   # (plain attributes of 8-bit layers: base.py:78-119 saves them by hand)
        lb.ind, lb.weight_cache = la.ind.clone(), (None if la.weight_cache is None else la.weight_cache.clone())
        lb.forward_without_precondition_len = la.forward_without_precondition_len
        lb._d.invalidate(outliers=True)
    y_b = b(norm_b(xs[3].clone().to(DEV)))
    assert torch.equal(y_b, y_a)
    assert b.up_proj_._wpk is None and b.up_proj_._buffers["q_weight"] is None       # ... and the block is back to one image


# ---------------------------------------------------------------------------------------------------------------
# W4A4 (int4 as FP6 codes): the same joint launch on the FP6 form of the kernel
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,n_out,bias", [(512, 11008, 4096, 128, False), (200, 584, 512, 70, True), (96, 96, 128, 0, False), (20, 1000, 1024, 129, False)])
def test_w4a4_one_launch_for_gate_and_up(M, N, K, n_out, bias):
    c = _case(M, N, K, n_out, bias, seed=M + N + n_out + 4, bit=4)
    u, g = c["up"], c["gate"]
    up_ref = O.linear_fused(c["qx"], u["qw"], c["sx"], u["sw"], xo=c["xo"], wo=u["wo"], addend=None, bias=u["bias"], act=0, bit=4)
    ref = O.linear_fused(c["qx"], g["qw"], c["sx"], g["sw"], xo=c["xo"], wo=g["wo"], addend=up_ref, bias=g["bias"], act=2, bit=4).astype(np.float32)
    y1 = _one_launch(c)
    y = n(y1).astype(np.float32)
    assert tuple(y1.shape) == (M, N) and np.isfinite(y).all()
    assert (np.abs(y - ref) <= ulp_tol(ref)).all(), float(np.abs(y - ref).max())
    y2 = _two_launches(c)
    assert torch.equal(y1, y2), int((y1 != y2).sum())
    if M * N > 2e6:
        return
    lib, names = _capi.load(), _capi.gemm_config_names()
    try:
        for nm in ("wr128x192_s16_d4_l2", "wr64x192_s16_d4_l2", "wr64x256_s16_d4_l2", "wr32x64_s8_d6_l1"):
            assert lib.mixq_gemm_set_config(names.index(nm)) == 0
            y3 = _one_launch(c)
            assert torch.equal(y3, y2), (nm, int((y3 != y2).sum()))
    finally:
        lib.mixq_gemm_set_config(-1)


def test_w4a4_mlp_block_on_the_joint_route_is_bit_identical():
    from mixq_amd import FasterTransformerRMSNorm, MixLibCache, MixLinear_GEMM, MixLlamaMLP, fused
    from mixq_amd import linear as L
    from mixq_amd import MixqConfig
    assert MixqConfig().pack_fmt4 == _capi.FMT_F6X128
    M, H, F = 64, 512, 1024
    outs = {}
    for joint in (False, True):
        torch.manual_seed(0)
        cache = MixLibCache(M, sigma=6, bit=4, device=DEV, config=MixqConfig(joint_gate_up=joint))
        cols = torch.randperm(H, generator=torch.Generator().manual_seed(1))[:8]
        ls = torch.ones(H); ls[cols] = 20.0
        mk = lambda k, nn_, sc: MixLinear_GEMM.from_linear(torch.nn.Linear(k, nn_, bias=False).half(), 4, cache=cache, layer_scales=sc, dev=DEV)
        lsd = torch.ones(F); lsd[:12] = 20.0
        gate, up, down = mk(H, F, ls), mk(H, F, ls), mk(F, H, lsd)
        norm = FasterTransformerRMSNorm((torch.rand(H) + 0.5).half().to(DEV), 1e-5, cache)
        norm.next_layer = up
        mlp = MixLlamaMLP(gate, down, up, cache)
        x = torch.randn(M, H, generator=torch.Generator().manual_seed(2)).half()
        x[:, cols] *= 20
        outs[joint] = [mlp(norm(x.clone().to(DEV))).clone() for _ in range(6)]
        if joint:
            assert mlp._joint is not None, "the joint route did not run"
            assert mixlib.fmt_of(mlp._joint["wpk"]) == _capi.FMT_F6X128 and up._wpk is None and gate._wpk is None
            assert mlp._joint["wpk"].numel() == 2 * F * H * 3 // 4          # 0.75 byte per weight, once
    for a, b in zip(outs[True], outs[False]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_joint_image_is_never_built_under_capture_and_a_graph_of_the_two_launch_route_survives_it():
    """The first frozen forward of a block happens under hipGraph capture: that graph takes the two-launch route (no packing kernels in
    the graph).  The next eager forward builds the joint image; the layers' own images, which the graph addresses, are kept alive for it
    - otherwise they are freed (one copy of the weights)."""
    from mixq_amd import fused
    M, H, F = 96, 512, 1536
    inner, norm, cache, xs = _block(M, H, F, False)
    up, gate = inner.up_proj_, inner.gate_proj_
    mlp = lambda x: inner(norm(x))
    prev = inner.config.joint_gate_up
    try:
        inner.config.joint_gate_up = False
        for x in xs[:3]:
            mlp(x.clone().to(DEV))
        y_ref = mlp(xs[3].clone().to(DEV))
        inner.config.joint_gate_up = True
        side = torch.cuda.Stream()
        xg = xs[3].clone().to(DEV)
        keep = xg.clone()
        with torch.cuda.stream(side):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                yg = mlp(xg)
        torch.cuda.synchronize()
        assert inner._joint is None and inner._two_launch_captured
        own = (up._wpk, gate._wpk)
        y = mlp(xs[3].clone().to(DEV))                                # eager: builds the joint image
        assert inner._joint is not None and up._wpk is None and torch.equal(y, y_ref)
        assert any(t is own[0] for t in up._d.retired) and any(t is own[1] for t in gate._d.retired)
        with torch.cuda.stream(side):
            for _ in range(2):
                xg.copy_(keep)
                gr.replay()
                torch.cuda.synchronize()
                assert torch.equal(yg, y_ref)
        # without a graph in the way the layers' own images are FREED by the build
        inner2, norm2, _, _ = _block(M, H, F, False, seed=3)
        for x in xs[:4]:
            inner2(norm2(x.clone().to(DEV)))
        assert inner2._joint is not None and not inner2.up_proj_._d.retired and not inner2.gate_proj_._d.retired
    finally:
        inner.config.joint_gate_up = prev


def test_moving_a_block_captured_on_the_joint_route_raises():
    """A graph captured through the joint route replays the joint image's raw address: .to(another device) must not free it silently."""
    M, H, F = 64, 256, 512
    inner, norm, cache, xs = _block(M, H, F, False)
    for x in xs[:4]:
        inner(norm(x.clone().to(DEV)))
    assert inner._joint is not None
    side = torch.cuda.Stream()
    xg = xs[4].clone().to(DEV)
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            inner(norm(xg))
    torch.cuda.synchronize()
    assert inner._joint.get("captured")
    with pytest.raises(RuntimeError, match="hipGraph capture"):
        inner.cpu()
    inner.to(DEV)                                                      # (same device: nothing moves, nothing raised)
    del gr
    inner.allow_move_after_capture = True
    for l in (inner.up_proj_, inner.gate_proj_, inner.down_proj_):
        l.allow_move_after_capture = True
    inner.cpu()
    assert not inner._joint["wpk"].is_cuda
    sd = inner.state_dict()                                            # the layers' rows come back out of the (host) joint image
    assert sd["up_proj_.q_weight"].shape == (F, H) and sd["gate_proj_.q_weight"].dtype == torch.int8


def test_random_shapes_one_launch_equals_two_launches():
    """Forty seeded random problems (ragged token counts, tiles hanging over N by every number of 16-column blocks, 1 .. 40 k-steps, 0 .. 200
    outlier columns, with and without bias, automatic and forced tilings): the joint launch against the two launches, bit for bit.  Random
    int8 operands straight from the generator - no oracle in the loop, so it runs in seconds."""
    rng = np.random.default_rng(2026)
    lib, names = _capi.load(), _capi.gemm_config_names()
    forms = [-1] + [names.index(nm) for nm in ("wr128x192_s16_d4_l2", "wr128x256_s16_d3_l2", "wr64x192_s16_d4_l2", "wr64x256_s16_d4_l2", "wr32x64_s8_d6_l1")]
    try:
        for case in range(40):
            M = int(rng.integers(1, 700))
            N = int(rng.integers(1, 160)) * 8                               # per layer; 2N % 16 == 0
            K = int(rng.integers(1, 41)) * 64
            n_out = int(rng.choice([0, 0, 1, 15, 16, 17, 33, 64, 65, 129, 200]))
            n_out = min(n_out, K)
            bias = bool(rng.integers(0, 2))
            g = torch.Generator().manual_seed(case)
            qx = mixlib.PackOperand(torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV), 1)
            sx = (torch.rand((M, 1), generator=g) * 0.02 + 0.001).half().to(DEV)
            pad = (n_out + 15) // 16 * 16
            xo = (torch.randn((M, max(pad, 16)), generator=g) * 3).half().to(DEV)[:, :n_out] if n_out else None
            lay = []
            for _ in range(2):
                qw = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
                sw = (torch.rand((1, N), generator=g) * 0.002 + 0.0001).half().to(DEV)
                wo = (torch.randn((N, max(pad, 16)), generator=g) * 0.05).half().to(DEV)[:, :n_out] if n_out else None
                b = torch.randn(N, generator=g).half().to(DEV) if bias else None
                lay.append((qw, sw, wo, b))
            (qu, su, wu, bu), (qg, sg, wg, bg) = lay
            up = mixlib.FusedLinear(qx, mixlib.PackOperand(qu, 2), sx, su, xo, wu, n_out, bu, M, N, K)
            ref = mixlib.FusedLinear(qx, mixlib.PackOperand(qg, 2), sx, sg, xo, wg, n_out, bg, M, N, K, act=_capi.ACT_SILU_MUL, addend=up)
            jw = mixlib.PackOperand(interleave_pair_rows(qu, qg), 2)
            jsw = interleave_pair_rows(su.reshape(-1), sg.reshape(-1)).reshape(1, -1)
            jwo = None
            if n_out:
                jwo = torch.zeros((2 * N, pad), dtype=torch.float16, device=DEV)
                jwo[:, :n_out] = interleave_pair_rows(wu, wg)
                jwo = jwo[:, :n_out]
            jb = interleave_pair_rows(bu, bg) if bias else None
            cfg = forms[case % len(forms)]
            assert lib.mixq_gemm_set_config(cfg) == 0
            y = mixlib.FusedLinear(qx, jw, sx, jsw, xo, jwo, n_out, jb, M, 2 * N, K, act=_capi.ACT_SILU_PAIR)
            lib.mixq_gemm_set_config(-1)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all() and torch.equal(y, ref), (case, M, N, K, n_out, bias, names[cfg] if cfg >= 0 else "auto", int((y != ref).sum()))
    finally:
        lib.mixq_gemm_set_config(-1)
