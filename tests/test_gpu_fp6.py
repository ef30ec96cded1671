"""W4A4 on the FP6 matrix pipe of the MI355X (MIXQ_FMT_F6X128, include/mixq_hip.h) through the C ABI: the storage format against a byte-level
restatement, the quantisers' direct output, the GEMM against the oracle and - bit for bit - against the int8-expansion W4A4 path,
every tiling that exists in the FP6 form, and the operator (MixLinear_GEMM) on top of it.  The int4 arithmetic is
/root/reference/mixquant/modules/linear.py:12-22 (packing), :209-226 (forward), :129-143 (from_linear); the carrier is this
framework's decision for a part without an int4 MFMA, so the bar is bit-exactness against the integer arithmetic."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mixq_amd import MixLibCache, MixLinear_GEMM, MixqConfig, _capi, mixlib  # noqa: E402
from mixq_amd import linear as L  # noqa: E402
from mixq_amd._capi import FMT_F6X128, FMT_R6X128, FMT_P16X64, FMT_PLAIN  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_parity import _fused_case, make_x, n, t, ulp_tol  # noqa: E402
from test_pack_properties import f6x128_reference, f6x128_unpack, r6x128_reference, r6x128_to_fragment_order  # noqa: E402

DEV = "cuda"
F6_TILINGS = ["wr128x192_s16_d4_l2", "wr128x128_s16_d4_l2", "wr64x128_s16_d4_l2", "wr64x192_s16_d4_l2", "wr64x256_s16_d4_l2",
              "wr32x64_s8_d6_l1"]                            # (round 4: the small-batch tiling has an FP6 form too - one weight image per layer)


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    assert "gfx950" in _capi.device_info()
    _capi.load().mixq_gemm_set_config(-1)
    yield
    _capi.load().mixq_gemm_set_config(-1)


def nibble_matrix(R, K, seed, lo=-8):
    rng = np.random.default_rng(seed)
    v = rng.integers(lo, 8, (R, K), dtype=np.int8)
    return v, O.pack_i4(v)


@pytest.mark.parametrize("R,K", [(1, 128), (16, 128), (37, 256), (100, 1024), (500, 4096)])
def test_pack_matches_the_format_byte_for_byte_and_unpack_inverts_it(R, K):
    v, p = nibble_matrix(R, K, seed=R + K)
    img = mixlib.PackOperand(t(p), FMT_F6X128)
    assert tuple(img.shape) == ((R + 15) // 16 * 16, K * 3 // 4) and mixlib.fmt_of(img) == FMT_F6X128
    want = f6x128_reference(v)
    assert np.array_equal(n(img).reshape(-1), want), "the device image differs from the format's restatement"
    assert np.array_equal(n(mixlib.UnpackOperand(img, R)), p)
    assert np.array_equal(f6x128_unpack(n(img).reshape(-1), R, K), v)
    # the host-side inverse the state_dict of a layer moved to the CPU uses
    assert np.array_equal(L._unpack_host(img.cpu(), R, FMT_F6X128).numpy(), p)
    # the activation side's row-contiguous form
    img_r = mixlib.PackOperand(t(p), FMT_R6X128)
    assert mixlib.fmt_of(img_r) == FMT_R6X128 and img_r.shape == img.shape
    assert np.array_equal(n(img_r).reshape(-1), r6x128_reference(v))
    assert np.array_equal(n(mixlib.UnpackOperand(img_r, R)), p)


def test_entry_points_validate_the_format():
    lib = _capi.load()
    one = torch.zeros(16, 64, dtype=torch.uint8, device=DEV)
    out = torch.zeros(16, 96, dtype=torch.uint8, device=DEV)
    assert lib.mixq_pack_operand(one.data_ptr(), out.data_ptr(), 16, 32, FMT_F6X128, None) == _capi.MIXQ_ESHAPE   # K = 64: not a whole block
    assert lib.mixq_gemm_i8_fused(one.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr(), None, 0, None, 0, 0, None, None, 0, None,
                                  out.data_ptr(), 64, 16, 64, 128, 0, _capi.XW_F6X128, None) == _capi.MIXQ_EINVAL   # an int4 layout
    xs = torch.zeros(16, dtype=torch.float16, device=DEV)
    x = torch.zeros(16, 128, dtype=torch.float16, device=DEV)
    assert lib.mixq_quant_fused(x.data_ptr(), None, 0, None, xs.data_ptr(), out.data_ptr(), None, None, 16, 128, 128, 0, 8, 6.0, FMT_R6X128,
                                None) == _capi.MIXQ_EINVAL                                                          # an int4 format


@pytest.mark.parametrize("M,K,ncols", [(512, 4096, 41), (37, 256, 0), (100, 11008 // 128 * 128, 110), (3, 128, 1)])
def test_quantisers_emit_the_format_directly(M, K, ncols):
    """mixq_quant_fused, mixq_find_row_scale and the fused RMSNorm with qfmt = F6X128 against the same kernels' plain nibble output
    (itself bit-exact against the oracle, test_gpu_parity.py): scales, outliers, zeroed x and every code of the valid rows."""
    rng = np.random.default_rng(M + K)
    ind_np = np.sort(rng.choice(K, ncols, replace=False)).astype(np.int32)
    x = make_x(M, K, seed=M, outlier_cols=ind_np)
    ind = t(ind_np) if ncols else None
    xs = [torch.zeros(M, dtype=torch.float16, device=DEV) for _ in range(2)]
    xa, xb = t(x.copy()), t(x.copy())
    q_plain, xo1 = mixlib.QuantFused(xa, ind, xs[0], 4, 6.0, fmt=FMT_PLAIN)
    xz = x.copy()
    if ncols:
        O.extract_outliers_zero(xz, ind_np)
    q_or, s_or = O.find_row_scale(xz, 4)
    for fmt in (FMT_R6X128, FMT_F6X128):                 # (the GEMM takes the first; the second is the weight format, supported for symmetry)
        xb = t(x.copy())
        xs[1].zero_()
        q_f6, xo2 = mixlib.QuantFused(xb, ind, xs[1], 4, 6.0, fmt=fmt)
        assert mixlib.fmt_of(q_f6) == fmt and q_f6.shape[1] == K * 3 // 4
        assert torch.equal(mixlib.UnpackOperand(q_f6, M), q_plain) and torch.equal(xs[0], xs[1]) and torch.equal(xa, xb)
        assert ncols == 0 or torch.equal(xo1, xo2)
        # ... the oracle's quantisation of the zeroed rows, code for code
        img = n(q_f6).reshape(-1)
        assert np.array_equal(f6x128_unpack(img if fmt == FMT_F6X128 else r6x128_to_fragment_order(img, M, K), M, K), O.unpack_i4_all(q_or))
        assert np.array_equal(n(xs[1]).view(np.uint16), s_or.view(np.uint16).reshape(-1))
        for cfg in (0, 4, 5, 6, 7, 8, 9):                # every launch geometry writes the same bytes (pairs of chunks per thread, or single chunks)
            assert _capi.load().mixq_quant_set_config(cfg) == 0
            q2, _ = mixlib.QuantFused(t(x.copy()), ind, xs[1], 4, 6.0, fmt=fmt)
            _capi.load().mixq_quant_set_config(-1)
            assert torch.equal(mixlib.UnpackOperand(q2, M), q_plain), cfg
    q_frs = mixlib.FindRowScalePacked(t(xz), xs[0], M, K, bit=4, fmt=FMT_R6X128)
    assert torch.equal(mixlib.UnpackOperand(q_frs, M), t(q_or))
    # fused RMSNorm + quantise
    w = t((rng.standard_normal(K) * 0.1 + 1).astype(np.float16))
    outs = []
    for fmt in (FMT_PLAIN, FMT_R6X128, FMT_F6X128):
        out = torch.empty(M, K, dtype=torch.float16, device=DEV)
        q, xo = mixlib.RMSNormQuantFused(t(x.copy()), w, out, 1e-5, ind, xs[0], 4, sigma=6.0, fmt=fmt)
        outs.append((out, q if fmt == FMT_PLAIN else mixlib.UnpackOperand(q, M), xo, xs[0].clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert (a is None and b is None) or torch.equal(a, b)


CASES = [
    # M, N, K, n_out, bias, addend, act
    (32, 96, 128, 0, False, False, 0),                  # one k-step
    (33, 100, 256, 3, True, False, 0),
    (200, 328, 512, 19, True, False, 1),
    (130, 200, 1024, 64, False, True, 0),               # exactly two tail k-steps
    (96, 320, 1024, 143, True, True, 2),                # the tail's loop form (> 64 outlier columns), gate * up epilogue
    (512, 1536, 4096, 41, False, False, 0),
    (64, 4096, 11008 // 128 * 128, 110, True, False, 0),  # 86 k-steps
]


@pytest.mark.parametrize("M,N,K,n_out,bias,addend,act", CASES)
def test_fp6_gemm_vs_oracle_and_bit_identical_to_the_int8_expansion(M, N, K, n_out, bias, addend, act):
    c = _fused_case(M, N, K, 4, seed=M + N + K + n_out, n_out=n_out, bias=bias, addend=addend or act == 2, act=act)
    ref = O.linear_fused(c["qx"], c["qw"], c["sx"], c["sw"], xo=c["xo"], wo=c["wo"], addend=c["addend"], bias=c["bias"], act=act,
                         bit=4).astype(np.float32)
    pad = (n_out + 15) // 16 * 16
    xo = wo = None
    if n_out:
        xo = torch.full((M, pad), float("nan"), dtype=torch.float16, device=DEV); xo[:, :n_out] = t(c["xo"]); xo = xo[:, :n_out]
        wo = torch.full((N, pad), float("nan"), dtype=torch.float16, device=DEV); wo[:, :n_out] = t(c["wo"]); wo = wo[:, :n_out]
    sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV); sx[:, 0] = t(c["sx"])
    args = dict(bit=4, act=act, addend=None if c["addend"] is None else t(c["addend"]))
    b = None if c["bias"] is None else t(c["bias"])
    run = lambda qx, qw: mixlib.FusedLinear(qx, qw, sx, t(c["sw"]), xo, wo, n_out, b, M, N, K, **args)
    lib, names = _capi.load(), _capi.gemm_config_names()
    y8 = run(mixlib.PackOperand(t(c["qx"]), FMT_P16X64), mixlib.PackOperand(t(c["qw"]), FMT_P16X64))
    x6, w6 = mixlib.PackOperand(t(c["qx"]), FMT_R6X128), mixlib.PackOperand(t(c["qw"]), FMT_F6X128)
    try:
        for cfg in [-1] + [names.index(nm) for nm in F6_TILINGS]:
            assert lib.mixq_gemm_set_config(cfg) == 0
            y6 = run(x6, w6)
            torch.cuda.synchronize()
            yn = n(y6).astype(np.float32)
            assert np.isfinite(yn).all()
            assert (np.abs(yn - ref) <= ulp_tol(ref)).all(), (names[cfg] if cfg >= 0 else "auto", float(np.abs(yn - ref).max()))
            assert torch.equal(y6, y8), (names[cfg] if cfg >= 0 else "auto", int((y6 != y8).sum()))
    finally:
        lib.mixq_gemm_set_config(-1)
    # a tiling without an FP6 form, and the layout with the wrong bit width, are refused - not silently served by something else
    assert lib.mixq_gemm_set_config(names.index("wr128x256_s16_d3_l2")) == 0
    try:
        with pytest.raises(_capi.MixqError):
            run(x6, w6)
    finally:
        lib.mixq_gemm_set_config(-1)
    with pytest.raises(RuntimeError):
        mixlib.FusedLinear(x6, mixlib.PackOperand(t(c["qw"]), FMT_P16X64), sx, t(c["sw"]), xo, wo, n_out, b, M, N, K, **args)
    with pytest.raises(RuntimeError):                     # activations in the WEIGHT format: the DMA would gather the wrong bytes
        mixlib.FusedLinear(mixlib.PackOperand(t(c["qx"]), FMT_F6X128), w6, sx, t(c["sw"]), xo, wo, n_out, b, M, N, K, **args)


def test_fp6_gemm_extreme_operands_stay_exact():
    """Activations of +-7 against weights of 7 / -8 over the longest K of the configured models (28672): the largest sums the accumulator
    sees (1.6e6 < 2^24)."""
    M, N, K = 48, 64, 28672
    for sign_w in (1, -1):
        qx = np.full((M, K), 7, dtype=np.int8); qw = np.full((N, K), 7 if sign_w > 0 else -8, dtype=np.int8)
        qx[1::2, ::3] = -7
        want = (qx.astype(np.int64) @ qw.astype(np.int64).T).astype(np.float64)
        sx = torch.ones((M, 1), dtype=torch.float16, device=DEV) * 2.0 ** -12
        sw = torch.ones((1, N), dtype=torch.float16, device=DEV) * 2.0 ** -10
        y = mixlib.FusedLinear(mixlib.PackOperand(t(O.pack_i4(qx)), FMT_R6X128), mixlib.PackOperand(t(O.pack_i4(qw)), FMT_F6X128), sx, sw, None, None, 0,
                               None, M, N, K, bit=4)
        ref = (want * 2.0 ** -22).astype(np.float16)
        assert np.array_equal(n(y).view(np.uint16), ref.view(np.uint16))


def _layer(M, K, N, ncols, bias, seed=0, **cfg):
    """(cfg: fields of the model's MixqConfig, e.g. pack_fmt4 = FMT_P16X64 for the nibble / int8-expansion path)"""
    torch.manual_seed(seed)
    lin = torch.nn.Linear(K, N, bias=bias).half()
    cols = torch.randperm(K, generator=torch.Generator().manual_seed(1))[:ncols]
    cache = MixLibCache(M, bit=4, device=DEV, config=MixqConfig(**cfg))
    scales = torch.ones(K) + torch.arange(K) * 1e-6
    scales[cols] = 20.0 + torch.arange(cols.numel()) * 1e-3
    layer = MixLinear_GEMM.from_linear(lin, 4, cache=cache, layer_scales=scales, dev=DEV)
    return layer, cache, cols


def test_four_bit_layer_runs_on_the_fp6_pipe_and_equals_the_int8_expansion():
    """MixLinear_GEMM(bit = 4): weights and activations travel as FP6 codes by default; warm-up (outlier search), frozen one-call forward,
    state_dict of the compacted layer, the host copy - all against the same layer on the nibble (int8-expansion) path."""
    M, K, N, ncols = 96, 1024, 320, 10
    outs = {}
    for fmt in (FMT_F6X128, FMT_P16X64):
        layer, cache, cols = _layer(M, K, N, ncols, True, pack_fmt4=fmt)
        ys = []
        for call in range(4):
            x = torch.randn(M, K, generator=torch.Generator().manual_seed(10 + call)).half()
            x[:, cols] *= 20
            ys.append(layer(x.to(DEV), None, True).clone())
        assert layer.add_outliers is False and layer._plan is not None
        xf = FMT_R6X128 if fmt == FMT_F6X128 else fmt
        assert mixlib.fmt_of(layer._packed_weight()) == fmt and layer.x_fmt() == xf and mixlib.fmt_of(cache.q_xcache) == xf
        outs[fmt] = (ys, layer, cache)
    for a, b in zip(outs[FMT_F6X128][0], outs[FMT_P16X64][0]):
        assert torch.equal(a, b)
    l6, l8 = outs[FMT_F6X128][1], outs[FMT_P16X64][1]
    assert l6._buffers["q_weight"] is None, "the frozen layer keeps only the packed image"
    sd6, sd8 = l6.state_dict(), l8.state_dict()
    assert torch.equal(sd6["q_weight"], sd8["q_weight"]) and sd6["q_weight"].shape == (N, K // 2)
    host = copy.deepcopy(l6).cpu()
    assert torch.equal(host.state_dict()["q_weight"], sd8["q_weight"].cpu())
    # against the oracle: the last forward's operands, recomputed on the host
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(13)).half()
    x[:, cols] *= 20
    xz = x.numpy().copy()
    ind = n(l6.ind).astype(np.int32)
    xo = O.extract_outliers_zero(xz, ind)
    qx, sx = O.find_row_scale(xz, 4)
    ref = O.linear_fused(qx, n(sd6["q_weight"]), sx, n(l6.scale_col), xo=xo, wo=n(l6.weight_cache), bias=n(l6.bias), bit=4).astype(np.float32)
    got = n(outs[FMT_F6X128][0][3]).astype(np.float32)
    assert (np.abs(got - ref) <= ulp_tol(ref)).all()


def test_reference_style_weights_hold_minus_eight_and_are_carried():
    """from_linear's 4-bit weights are clamp(round(w / (rowmax / 10)), -8, 7) (linear.py:136-139): -8 is common, and an FP6 E3M2 value."""
    layer, cache, cols = _layer(64, 512, 128, 6, False)
    qw = n(layer.q_weight)
    assert ((qw & 0xF) == 8).any() or ((qw >> 4) == 8).any(), "expected -8 among reference-style 4-bit weights"
    wpk = layer._packed_weight()
    assert mixlib.fmt_of(wpk) == FMT_F6X128
    assert np.array_equal(n(mixlib.UnpackOperand(wpk, 128)), qw)


def test_mlp_block_w4a4_shares_the_fp6_activation():
    """up_proj / gate_proj of the fused MLP read ONE quantised activation (mlp.py:37-70): with 4-bit layers it is an FP6 image written
    by the fused RMSNorm, and the block equals the nibble-path block bit for bit."""
    from mixq_amd import FasterTransformerRMSNorm, MixLlamaMLP
    M, H, F = 64, 512, 1024
    res = {}
    for fmt in (FMT_F6X128, FMT_P16X64):
        torch.manual_seed(0)
        cache = MixLibCache(M, sigma=6, bit=4, device=DEV, config=MixqConfig(pack_fmt4=fmt))
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
        ys = [mlp(norm(x.clone().to(DEV))).clone() for _ in range(4)]
        assert mixlib.fmt_of(up._packed_weight()) == fmt
        res[fmt] = ys
    for a, b in zip(res[FMT_F6X128], res[FMT_P16X64]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_a_four_bit_layer_keeps_one_weight_image_by_default():
    """Round 4 (VERDICT r03 #5b): a 4-bit layer holds ONE resident weight image - the FP6 one, 0.75 byte per weight - whatever batch sizes
    it sees: small batches run the FP6 form of the 32 x 64 weight-stream tiling on it.  Same bits as the nibble-only layer."""
    assert MixqConfig().small_batch_m4 == 0, "the second (nibble) image is opt-in"
    K, N, ncols = 1024, 320, 10
    outs = {}
    for fmt in (FMT_F6X128, FMT_P16X64):
        layer, cache, cols = _layer(96, K, N, ncols, True, pack_fmt4=fmt)
        ys = []
        for call, M in enumerate((96, 96, 96, 16, 96, 1, 32, 33)):
            x = torch.randn(M, K, generator=torch.Generator().manual_seed(40 + call)).half()
            x[:, cols] *= 20
            ys.append(layer(x.to(DEV), None, True).clone())
            if fmt == FMT_F6X128:
                assert mixlib.fmt_of(cache.q_xcache) == FMT_R6X128 and layer._wpk_small is None, M
        if fmt == FMT_F6X128:
            assert layer._buffers["q_weight"] is None and mixlib.fmt_of(layer._wpk) == FMT_F6X128
            assert layer._wpk.numel() == N * K * 3 // 4, "resident bytes of the layer: 0.75 per weight"
        outs[fmt] = ys
    for a, b in zip(outs[FMT_F6X128], outs[FMT_P16X64]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_small_batches_of_a_four_bit_layer_stream_a_nibble_image():
    """Opt-in (MixqConfig.small_batch_m4 = 32; narrow layers at decode: 9-10 vs 11-12 us at 4096 -> 4096, profiles/r04_w4a4_small_batch.txt): M <= 32 is a
    weight stream - the layer serves it from a second, nibble image (built when the first small batch arrives, also after the plain matrix
    was dropped) with P16X64 activations; larger batches keep the FP6 pair.  Same bits either way."""
    K, N, ncols = 1024, 320, 10
    outs = {}
    for fmt in (FMT_F6X128, FMT_P16X64):
        layer, cache, cols = _layer(96, K, N, ncols, True, pack_fmt4=fmt, small_batch_m4=32)
        ys = []
        for call, M in enumerate((96, 96, 96, 16, 96, 1, 32, 33)):
            x = torch.randn(M, K, generator=torch.Generator().manual_seed(40 + call)).half()
            x[:, cols] *= 20
            if fmt == FMT_F6X128 and call == 3:
                assert layer._wpk_small is None and layer._buffers["q_weight"] is None       # frozen, compacted, no small batch seen yet
            ys.append(layer(x.to(DEV), None, True).clone())
            if fmt == FMT_F6X128:
                small = M <= 32
                assert mixlib.fmt_of(cache.q_xcache) == (FMT_P16X64 if small else FMT_R6X128), (M, mixlib.fmt_of(cache.q_xcache))
        if fmt == FMT_F6X128:
            assert layer._wpk_small is not None and mixlib.fmt_of(layer._wpk_small) == FMT_P16X64 and mixlib.fmt_of(layer._wpk) == FMT_F6X128
        outs[fmt] = ys
    for a, b in zip(outs[FMT_F6X128], outs[FMT_P16X64]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("M,K,N,ncols", [(37, 384, 100, 3), (130, 128, 72, 5), (64, 4224, 136, 12), (40, 640, 16, 1), (96, 1152, 1000, 0)])
def test_four_bit_layer_with_ragged_k_and_n(M, K, N, ncols):
    """Odd counts of 128-column FP6 blocks, out_features that are not a multiple of 16, batches that are not a multiple of 16: the
    zero-padded FP6 pair gives the nibble path's bits, warm-up and frozen calls alike."""
    outs = {}
    for fmt in (FMT_F6X128, FMT_P16X64):
        layer, cache, cols = _layer(M, K, N, ncols, True, seed=3, pack_fmt4=fmt)
        ys = []
        for call in range(4):
            x = torch.randn(M, K, generator=torch.Generator().manual_seed(70 + call)).half()
            x[:, cols] *= 20
            ys.append(layer(x.to(DEV), None, True).clone())
        assert mixlib.fmt_of(layer._packed_weight()) == fmt
        outs[fmt] = (ys, layer)
    for a, b in zip(outs[FMT_F6X128][0], outs[FMT_P16X64][0]):
        assert a.shape == (M, N) and torch.isfinite(a).all() and torch.equal(a, b)
    assert torch.equal(outs[FMT_F6X128][1].state_dict()["q_weight"], outs[FMT_P16X64][1].state_dict()["q_weight"])


def test_unsupported_layer_shapes_are_refused_when_the_layer_is_built():
    """The kernels' shape contract (K % 128 for 4-bit operands, K % 64 for int8, N % 4: include/mixq_hip.h) is stated by the constructor."""
    for bit, K, N in [(4, 192, 64), (4, 200, 64), (8, 96, 64), (8, 128, 66)]:
        with pytest.raises(ValueError, match="in_features"):
            MixLinear_GEMM(K, N, False, DEV, bit, cache=MixLibCache(16, bit=bit, device=DEV))
    MixLinear_GEMM(192, 64, False, DEV, 8, cache=MixLibCache(16, bit=8, device=DEV))


@pytest.mark.parametrize("M,N,K,n_out", [(512, 11008, 4096, 128), (512, 12288, 4096, 128), (512, 4096, 4096, 128), (2048, 11008, 4096, 128), (512, 28672, 8192, 128)])
def test_fp6_gemm_at_the_baseline_shapes_equals_the_nibble_path_bit_for_bit(M, N, K, n_out):
    """BASELINE config 2's layers (and a prefill batch, and the 70b width) at full size: too big for the oracle in a test, so the property
    checked is the one the carrier promises - the FP6 pipe returns the int8-expansion path's bits - plus a sampled-rows check against
    exact integer arithmetic on the host."""
    rng = np.random.default_rng(M + N + K)
    vx = rng.integers(-7, 8, (M, K), dtype=np.int8); vw = rng.integers(-8, 8, (N, K), dtype=np.int8)
    qx, qw = t(O.pack_i4(vx)), t(O.pack_i4(vw))
    sx = t((rng.random((M, 1)) * 0.01 + 0.001).astype(np.float16)); sw = t((rng.random((1, N)) * 0.01 + 0.001).astype(np.float16))
    pad = (n_out + 15) // 16 * 16
    xo = torch.randn((M, pad), device=DEV, dtype=torch.float16)[:, :n_out]; wo = torch.randn((N, pad), device=DEV, dtype=torch.float16)[:, :n_out]
    bias = torch.randn(N, device=DEV, dtype=torch.float16)
    want = mixlib.FusedLinear(mixlib.PackOperand(qx, FMT_P16X64), mixlib.PackOperand(qw, FMT_P16X64), sx, sw, xo, wo, n_out, bias, M, N, K, bit=4)
    got = mixlib.FusedLinear(mixlib.PackOperand(qx, FMT_R6X128), mixlib.PackOperand(qw, FMT_F6X128), sx, sw, xo, wo, n_out, bias, M, N, K, bit=4)
    assert torch.equal(want, got)
    rows = rng.choice(M, 6, replace=False)
    acc = vx[rows].astype(np.int64) @ vw.astype(np.int64).T
    ref = acc.astype(np.float64) * n(sx)[rows].astype(np.float64) * n(sw).astype(np.float64) \
        + n(xo)[rows].astype(np.float64) @ n(wo).astype(np.float64).T + n(bias).astype(np.float64)
    assert (np.abs(n(got)[rows].astype(np.float64) - ref) <= ulp_tol(ref.astype(np.float32))).all()
