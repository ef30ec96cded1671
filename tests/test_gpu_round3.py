"""Round-3 parity tests on the MI355X (through the C ABI): the one-call forward of a frozen layer against the two-call route, the
operator-level full-size forwards round 2 covered only by the int32 checksum, and the bench line's other configurations."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mixq_amd import MixLibCache, MixLinear_GEMM, _capi, mixlib  # noqa: E402
from mixq_amd import linear as L  # noqa: E402
from oracle import oracle as O  # noqa: E402

DEV = "cuda"
GATE = 1e-2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def n(x):
    return x.detach().cpu().numpy()


def ulp_tol(ref):
    a = np.abs(ref.astype(np.float32))
    ulp = np.where(a > 0, 2.0 ** (np.floor(np.log2(np.maximum(a, 6e-5))) - 10), 2.0 ** -24)
    return np.maximum(2 * ulp, 1e-3)


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    assert "gfx950" in _capi.device_info()
    _capi.load().mixq_gemm_set_config(-1)
    yield
    _capi.load().mixq_gemm_set_config(-1)


def frozen_layer(M, K, N, bit, ncols, bias, seed=0):
    torch.manual_seed(seed)
    lin = torch.nn.Linear(K, N, bias=bias).half()
    cols = torch.randperm(K, generator=torch.Generator().manual_seed(1))[:ncols]
    cache = MixLibCache(M, bit=bit, device=DEV)
    if bit == 8:
        layer = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=DEV)
    else:
        scales = torch.ones(K) + torch.arange(K) * 1e-6
        scales[cols] = 20.0 + torch.arange(cols.numel()) * 1e-3
        layer = MixLinear_GEMM.from_linear(lin, 4, cache=cache, layer_scales=scales, dev=DEV)
    for call in range(3):
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(10 + call)).half()
        x[:, cols] *= 20
        layer(x.to(DEV), None, True)
    assert layer.add_outliers is False
    return layer, cache, cols


@pytest.mark.parametrize("M,K,N,bit,ncols,bias", [(512, 4096, 11008, 8, 41, False), (96, 1024, 320, 8, 5, True), (40, 512, 256, 8, 0, False),
                                                   (33, 1024, 192, 8, 16, True), (64, 1024, 256, 4, 10, False), (512, 4096, 4096, 4, 41, True)])
def test_one_call_forward_is_bit_identical(M, K, N, bit, ncols, bias):
    """mixq_linear_forward (one C call, kept argument block) against the two-call route (mixq_quant_fused, mixq_gemm_i*_fused) on
    the same frozen layer and input: y, x_scale, q_x, the extracted outliers and the zeroed x must agree bit for bit."""
    layer, cache, cols = frozen_layer(M, K, N, bit, ncols, bias)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(77)).half()
    x[:, cols] *= 20
    res = {}
    for one in (False, True, True):                                  # the third run re-uses the kept plan
        layer.config.one_call_forward = one
        xd = x.to(DEV)
        cache.x_scale.zero_()
        y = layer(xd, None, True)
        torch.cuda.synchronize()
        if one:
            assert layer._plan is not None, "the frozen layer did not take the one-call route"
        qplain = cache.q_xcache if mixlib.fmt_of(cache.q_xcache) == 0 else mixlib.UnpackOperand(cache.q_xcache, M)   # (pad rows of a packed image are never written)
        got = (n(y).copy(), n(cache.x_scale[:M]).copy(), n(qplain).copy(), n(xd).copy(),
               None if cache.activation_outliers is None else n(cache.activation_outliers).copy(), mixlib.fmt_of(cache.q_xcache))
        if not res:
            res = got
            continue
        for a, b in zip(res[:4], got[:4]):
            assert np.array_equal(a.view(np.uint8) if a.dtype != np.uint8 else a, b.view(np.uint8) if b.dtype != np.uint8 else b)
        assert (res[4] is None) == (got[4] is None) and (res[4] is None or np.array_equal(res[4].view(np.uint16), got[4].view(np.uint16)))
        assert res[5] == got[5]
    layer.config.one_call_forward = True
    # and the route is the operator's real arithmetic: sampled rows against the oracle
    rows = sorted({0, M // 2, M - 1})
    xh = x.numpy()[rows].copy()
    ind = n(layer.ind).astype(np.int32)
    xo = O.extract_outliers_zero(xh, ind) if ind.size else None
    qx, sx = O.find_row_scale(xh, bit)
    ref = O.linear_fused(qx, n(layer.q_weight), sx, n(layer.scale_col), xo=xo, wo=(n(layer.weight_cache) if ind.size else None),
                         bias=(n(layer.bias) if bias else None), bit=bit).astype(np.float32)
    assert (np.abs(res[0][rows].astype(np.float32) - ref) <= ulp_tol(ref)).all()


def test_one_call_plan_follows_the_layer_state():
    """The kept argument block is dropped when anything it carries an address of is replaced: another batch size, a strided input, a
    reloaded state dict, an in-place rewrite of `ind`'s companion tensors."""
    layer, cache, cols = frozen_layer(64, 1024, 256, 8, 7, True)
    x = torch.randn(64, 1024, generator=torch.Generator().manual_seed(5)).half()
    x[:, cols] *= 20
    y1 = layer(x.to(DEV), None, True)
    p1 = layer._plan
    y1b = layer(x.to(DEV), None, True)
    assert layer._plan is p1 and torch.equal(y1, y1b)
    y2 = layer(x[:32].to(DEV), None, True)                              # another M: a new block
    assert layer._plan is not p1 and torch.equal(y2, y1[:32])
    wide = torch.zeros(64, 2048, dtype=torch.float16, device=DEV)       # a row-strided view of a wider buffer
    wide[:, :1024] = x.to(DEV)
    y3 = layer(wide[:, :1024], None, True)
    assert torch.equal(y3, y1)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    sd["bias"] = sd["bias"] + 1
    layer.load_state_dict(sd)
    y4 = layer(x.to(DEV), None, True)
    layer.config.one_call_forward = False
    y4b = layer(x.to(DEV), None, True)
    layer.config.one_call_forward = True
    assert torch.equal(y4, y4b) and not torch.equal(y4, y1)


def test_linear_forward_c_entry_validates_its_block():
    lib = _capi.load()
    assert lib.mixq_linear_forward(None, None) == _capi.MIXQ_EINVAL
    a = _capi.LinearArgs()
    a.bit = 5
    assert lib.mixq_linear_forward(C.byref(a), None) == _capi.MIXQ_EINVAL
    a.bit, a.wfmt, a.qfmt = 8, _capi.FMT_F16X64, _capi.FMT_PLAIN          # fragment-order weights need P16X64 activations
    assert lib.mixq_linear_forward(C.byref(a), None) == _capi.MIXQ_EINVAL
    a.wfmt, a.qfmt, a.n_cap = _capi.FMT_PLAIN, _capi.FMT_PLAIN, 16          # outlier capacity without the tail's operands
    assert lib.mixq_linear_forward(C.byref(a), None) == _capi.MIXQ_EINVAL


# ---------------------------------------------------------------------------------------------------------------
# operator-level forwards at the BASELINE shapes round 2 covered only through the raw int32 checksum
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,N", [(28672, 8192), (11008, 4096), (4096, 6144), (8192, 10240), (4096, 4096), (4096, 12288)])
def test_full_size_operator_remaining_shapes_with_one_percent_outliers(K, N):
    """M = 512, 1 % outlier columns (287 at K = 28672: the > 128 tail at full size), prediction frozen; quantise + GEMM + fp16 tail +
    dequant epilogue on sampled rows against the oracle (<= 2 fp16 ulp) and the north-star gate."""
    M = 512
    layer, cache, cols = frozen_layer(M, K, N, 8, round(0.01 * K), False)
    assert set(cols.tolist()) <= set(n(layer.ind).tolist())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(12)).half()
    x[:, cols] *= 20
    y = layer(x.to(DEV), None, True)
    rows = [0, 255, 511]
    xh = x.numpy()[rows].copy()
    ind = n(layer.ind).astype(np.int32)
    xo = O.extract_outliers_zero(xh, ind)
    qx, sx = O.find_row_scale(xh, 8)
    qw, sw, wo = n(layer.q_weight), n(layer.scale_col), n(layer.weight_cache)
    ref = O.linear_fused(qx, qw, sx, sw, xo=xo, wo=wo).astype(np.float32)
    got = n(y)[rows].astype(np.float32)
    assert (np.abs(got - ref) <= ulp_tol(ref)).all(), float(np.abs(got - ref).max())
    gate = O.linear_dequant_ref(qx, qw, sx, sw, xo=xo, ind=ind, wo=wo)
    a = np.abs(gate)
    assert (np.abs(got - gate) <= np.maximum(GATE, 2.0 ** (np.floor(np.log2(np.maximum(a, 1.0))) - 10))).all()


# ---------------------------------------------------------------------------------------------------------------
# bench.py's other configurations work first time (config 3's shape for the SCALE command, config 2's bit width)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("extra,K,N,bit", [(["--shape", "8192,28672"], 8192, 28672, 8), (["--bit", "4"], 4096, 11008, 4)])
def test_bench_other_configs_run(extra, K, N, bit):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"] + extra,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["config"]["K"] == K and out["config"]["N"] == N and out["steps"] == 5
    assert out["value"] == pytest.approx(2.0 * 512 * K * N / (out["ms_per_step"] * 1e-3) / 1e12, rel=1e-3)
    assert out["max_abs_err_vs_dequant_linear"] <= (1e-2 if bit == 8 else 4e-2)       # (W4: |y| reaches 30 at 128 fp16 columns: 1 ulp = 1.6e-2)
    assert f"W{bit}A{bit}O16" in out["metric"] and out["roofline"]["frac"] > 0.1


def test_compacted_layer_state_dict_edge_cases():
    """ADVICE r02 (linear.py compacted weights): a strict load without q_weight reports the missing key; a q_weight of the wrong
    shape is an error that leaves the packed weights in place; state_dict() works after the module moved to the host."""
    layer, cache, cols = frozen_layer(32, 512, 384, 8, 3, True)
    assert layer._buffers["q_weight"] is None and layer._wpk is not None
    x = torch.randn(32, 512, generator=torch.Generator().manual_seed(9)).half()
    x[:, cols] *= 20
    y0 = layer(x.to(DEV), None, True)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    with pytest.raises(RuntimeError, match="Missing key"):
        layer.load_state_dict({k: v for k, v in sd.items() if k != "q_weight"})
    bad = dict(sd)
    bad["q_weight"] = sd["q_weight"][:, :256].contiguous()
    with pytest.raises(RuntimeError, match="size mismatch"):
        layer.load_state_dict(bad)
    assert layer._wpk is not None and torch.equal(layer(x.to(DEV), None, True), y0)
    layer.cpu()
    host_sd = layer.state_dict()
    assert not host_sd["q_weight"].is_cuda and torch.equal(host_sd["q_weight"], sd["q_weight"].cpu())
    layer.to(DEV)
    cache2 = MixLibCache(32, device=DEV)
    layer.cache = cache2
    assert torch.equal(layer(x.to(DEV), cache2, True), y0)


# ---------------------------------------------------------------------------------------------------------------
# down_proj's pre-pass maximum as a side output of gate_proj's GEMM (mlp.py:57-70 + linear.py:187-193)
# ---------------------------------------------------------------------------------------------------------------
def _mask_words(K, cols, dev=DEV):
    bits = np.zeros(((K + 31) // 32) * 32, dtype=np.uint64)
    bits[list(cols)] = 1
    w = (bits.reshape(-1, 32) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    return torch.from_numpy(w.view(np.int32).copy()).to(dev)


@pytest.mark.parametrize("M,N,K,act,bias,masked", [(100, 260, 512, 0, True, 7), (512, 1536, 1024, 2, False, 41), (33, 64, 128, 1, True, 0),
                                                     (257, 1000, 256, 2, True, 110), (16, 8192, 256, 0, False, 5)])
def test_gemm_row_amax_side_output_is_the_exact_masked_maximum(M, N, K, act, bias, masked):
    """mixq_gemm_i8_fused_amax on every weights-in-registers tiling: y is bit-identical to the plain entry point and row_amax[m] is
    exactly max over the unmasked columns of the fp16 bit patterns |y[m, :]| (integer atomics: order-independent)."""
    rng = np.random.default_rng(M + N + K)
    qx = torch.from_numpy(rng.integers(-127, 128, size=(M, K), dtype=np.int8)).to(DEV)
    qw = torch.from_numpy(rng.integers(-127, 128, size=(N, K), dtype=np.int8)).to(DEV)
    sx = (torch.rand(M, 1) * 0.01 + 0.001).half().to(DEV)
    sw = (torch.rand(1, N) * 0.01 + 0.001).half().to(DEV)
    b = torch.randn(N).half().to(DEV) if bias else None
    add = torch.randn(M, N).half().to(DEV) if act == 2 else None
    cols = sorted(rng.choice(N, size=masked, replace=False).tolist()) if masked else []
    mask = _mask_words(N, cols) if masked else None
    xp, wp = mixlib.PackOperand(qx, 1), mixlib.PackOperand(qw, 2)
    lib = _capi.load()
    names = _capi.gemm_config_names()
    try:
        for cfg in [-1] + [i for i, nm in enumerate(names) if nm.startswith("wr") and not nm.endswith(("_k2", "_pair"))]:      # (split-K: own tests, needs a workspace)
            assert lib.mixq_gemm_set_config(cfg) == 0
            if not mixlib.amax_supported(M, N, K, 1, 2) and cfg == -1:
                continue
            y0 = mixlib.FusedLinear(xp, wp, sx, sw, None, None, 0, b, M, N, K, act=act, addend=add)
            buf = torch.zeros(M + 3, dtype=torch.int32, device=DEV)
            y1 = mixlib.FusedLinear(xp, wp, sx, sw, None, None, 0, b, M, N, K, act=act, addend=add, row_amax=buf, col_mask=mask)
            assert torch.equal(y0, y1), names[cfg]
            bits = n(y1).view(np.uint16).astype(np.int64) & 0x7fff
            if cols:
                bits[:, cols] = 0
            assert np.array_equal(n(buf)[:M].astype(np.int64), bits.max(axis=1)), names[cfg]
            assert (n(buf)[M:] == 0).all()
    finally:
        lib.mixq_gemm_set_config(-1)


@pytest.mark.parametrize("M,K,ncols,bit,fmt", [(64, 4096, 41, 8, 1), (33, 11008, 110, 8, 1), (16, 256, 0, 8, 0), (40, 1024, 130, 4, 1), (7, 28672, 287, 8, 1)])
def test_known_maximum_quantiser_writes_the_same_bytes(M, K, ncols, bit, fmt):
    """mixq_quant_known_amax (one pass, maxima handed over) against mixq_quant_fused (two passes): q, x_scale, the extracted outliers and
    the zeroed x agree bit for bit, and the maxima buffer comes back cleared."""
    rng = np.random.default_rng(K + ncols)
    x = rng.standard_normal((M, K)).astype(np.float16)
    cols = np.sort(rng.choice(K, size=ncols, replace=False)).astype(np.int32)
    x[:, cols] *= 20
    x[M // 2] = 0                                                       # an all-zero row: scale 0, q 0
    ind = torch.from_numpy(cols).to(DEV) if ncols else None
    xa, xb = torch.from_numpy(x).to(DEV), torch.from_numpy(x).to(DEV)
    sa, sb = torch.zeros(M, 1, dtype=torch.float16, device=DEV), torch.zeros(M, 1, dtype=torch.float16, device=DEV)
    qa, xoa = mixlib.QuantFused(xa, ind, sa, bit, 6.0, fmt=fmt)
    xm = np.abs(x.astype(np.float32))
    if ncols:
        xm[:, cols] = 0
    amax = torch.from_numpy(xm.max(axis=1).astype(np.float16).view(np.uint16).astype(np.int32)).to(DEV)
    mask = L.kept_outlier_map(ind, K) if ncols else None               # the layer's kept outlier map: bits, count, AND-masks (include/mixq_hip.h)
    qb = torch.empty_like(qa)
    ldo = (ncols + 15) // 16 * 16
    xob = torch.empty((M, ldo), dtype=torch.float16, device=DEV) if ncols else None
    _capi.call("mixq_quant_known_amax", xb.data_ptr(), None if ind is None else ind.data_ptr(), ncols, None, amax.data_ptr(),
               None if mask is None else mask.data_ptr(), 0 if mask is None else mask.numel(), sb.data_ptr(), qb.data_ptr(), None if xob is None else xob.data_ptr(), None,
               M, K, K, ldo, bit, 6.0, fmt, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(sa, sb) and torch.equal(xa, xb) and (n(amax) == 0).all()
    ua = mixlib.UnpackOperand(mixlib.set_fmt(qa, fmt), M, fmt) if fmt else qa
    ub = mixlib.UnpackOperand(mixlib.set_fmt(qb, fmt), M, fmt) if fmt else qb
    assert torch.equal(ua, ub)
    if ncols:
        assert torch.equal(xoa, xob[:, :ncols])


def test_mlp_block_with_the_fused_row_maximum_is_bit_identical():
    """MixLlamaMLP with down_proj's pre-pass maximum taken from gate_proj's epilogue against the two-pass quantiser: same output bits,
    same down_proj x_scale, the hand-over buffer is consumed (cleared) every forward, also under hipGraph replay."""
    from mixq_amd import FasterTransformerRMSNorm, MixLlamaMLP, fused
    M, H, F = 96, 512, 1536
    torch.manual_seed(0)
    cache = MixLibCache(M, device=DEV)
    mk = lambda k, nn_: MixLinear_GEMM.from_linear(torch.nn.Linear(k, nn_, bias=True).half(), 8, cache=cache, dev=DEV)
    gate, up, down = mk(H, F), mk(H, F), mk(F, H)
    inner = MixLlamaMLP(gate, down, up, cache)
    norm = FasterTransformerRMSNorm((torch.rand(H) + 0.5).half().to(DEV), 1e-5, cache)      # fills the cache for up_proj (norm.py:24-33)
    norm.next_layer = up
    mlp = lambda x: inner(norm(x))
    g = torch.Generator().manual_seed(1)
    cols = torch.randperm(H, generator=g)[:5]
    xs = []
    for c in range(4):
        x = torch.randn(M, H, generator=g).half()
        x[:, cols] *= 20
        xs.append(x)
    prev = inner.config.fuse_down_amax
    try:
        inner.config.fuse_down_amax = False
        for x in xs[:3]:
            mlp(x.clone().to(DEV))                                    # freeze every layer's outlier search
        assert not down.add_outliers and not up.add_outliers
        y_ref = mlp(xs[3].clone().to(DEV))
        sx_ref = cache.x_scale[:M].clone()
        inner.config.fuse_down_amax = True
        y = mlp(xs[3].clone().to(DEV))
        assert down._amax_buf is not None and not down._amax_dirty, "down_proj did not take the hand-over"
        assert torch.equal(y, y_ref) and torch.equal(cache.x_scale[:M], sx_ref)
        assert int(down._amax_buf.abs().sum()) == 0
        side = torch.cuda.Stream()
        xg = xs[3].clone().to(DEV)
        keep = xg.clone()
        with torch.cuda.stream(side):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                yg = mlp(xg)
            for _ in range(3):
                xg.copy_(keep)
                gr.replay()
                torch.cuda.synchronize()
                assert torch.equal(yg, y_ref)
    finally:
        inner.config.fuse_down_amax = prev


# ---------------------------------------------------------------------------------------------------------------
# the LDS-staged tilings read the operator's ONE weight image (fragment order) too; prefill batches route to the 256 x 256 one
# ---------------------------------------------------------------------------------------------------------------
def test_lds_staged_tilings_read_fragment_order_weights_bit_identically():
    """Every tiling of gemm.hip with the weights in MIXQ_FMT_F16X64 (remapped DMA source) against the same tiling with P16X64 weights:
    int8 and int4, ragged shapes, outlier tail, addend / SiLU / bias - identical bits (it is the same LDS image)."""
    from test_gpu_parity import _fused_case, _run_fused, _tiled_configs, bits
    lib = _capi.load()
    names = _capi.gemm_config_names()
    cases = [(100, 260, 512, 8, 17, True, True, 1), (257, 1000, 1024, 8, 41, False, False, 0), (33, 36, 320, 8, 3, True, False, 2),
             (64, 128, 1024, 4, 128, False, False, 0), (130, 200, 512, 4, 16, True, True, 1), (512, 384, 256, 8, 0, False, False, 0)]
    try:
        for (M, N, K, bit, n_out, bias, addend, act) in cases:
            c = _fused_case(M, N, K, bit, seed=M + N + K, n_out=n_out, bias=bias, addend=addend or act == 2, act=act)
            for cfg in _tiled_configs():
                assert lib.mixq_gemm_set_config(cfg) == 0
                y1 = n(_run_fused(c, 1))
                y2 = n(_run_fused(c, 2))
                assert np.array_equal(bits(y1), bits(y2)), (names[cfg], M, N, K, bit)
    finally:
        lib.mixq_gemm_set_config(-1)


def test_prefill_batches_stay_on_the_weights_in_registers_tilings_and_stay_exact():
    """4096 tokens x 4096 -> 4096 with fragment-order weights: the automatic choice is a weights-in-registers tiling (round 5: 128 x 256 is
    ahead of the LDS-staged 256 x 256 tiling at every prefill shape measured, profiles/r05_prefill_sweep.txt), the product is exact against
    the integer reference, and the operator agrees with the oracle on sampled rows."""
    lib = _capi.load()
    names = _capi.gemm_config_names()
    M, N, K = 4096, 4096, 4096
    assert names[lib.mixq_gemm_pick_config_fmt(M, N, K, 8, 2)].startswith("wr128x")
    assert names[lib.mixq_gemm_pick_config_fmt(2048, 11008, 4096, 8, 2)].startswith("wr128x256")
    assert lib.mixq_gemm_amax_supported(M, N, K, _capi.X_PACKED | _capi.W_F16X64) == 1
    g = torch.Generator().manual_seed(3)
    qx = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    qw = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
    sx = torch.full((M, 1), 2.0 ** -8, dtype=torch.float16, device=DEV)
    sw = torch.full((1, N), 2.0 ** -8, dtype=torch.float16, device=DEV)
    y = mixlib.FusedLinear(mixlib.PackOperand(qx, 1), mixlib.PackOperand(qw, 2), sx, sw, None, None, 0, None, M, N, K)
    rows = torch.tensor([0, 1, 255, 256, 2047, 4095], device=DEV)
    want = ((qx[rows].double() @ qw.double().T) * 2.0 ** -16).to(torch.float16)
    assert torch.equal(y[rows], want)
    layer, cache, cols = frozen_layer(M, K, N, 8, 41, False)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(12)).half()
    x[:, cols] *= 20
    yo = layer(x.to(DEV), None, True)
    rs = [0, 2047, 4095]
    xh = x.numpy()[rs].copy()
    ind = n(layer.ind).astype(np.int32)
    xo = O.extract_outliers_zero(xh, ind)
    qxo, sxo = O.find_row_scale(xh, 8)
    ref = O.linear_fused(qxo, n(layer.q_weight), sxo, n(layer.scale_col), xo=xo, wo=n(layer.weight_cache)).astype(np.float32)
    assert (np.abs(n(yo)[rs].astype(np.float32) - ref) <= ulp_tol(ref)).all()


def test_row_maximum_hand_over_is_dropped_when_the_activation_was_edited():
    """The maxima gate_proj's GEMM leaves for down_proj describe the tensor as the GEMM wrote it: an in-place edit in between (what the
    reference's own `gate_output *= up_output` would be) must send down_proj back to the two-pass quantiser - and the stale buffer is
    cleaned before the next producer run."""
    from mixq_amd import FasterTransformerRMSNorm
    M, H, F = 64, 512, 1024
    torch.manual_seed(0)
    cache = MixLibCache(M, device=DEV)
    mk = lambda k, nn_: MixLinear_GEMM.from_linear(torch.nn.Linear(k, nn_, bias=False).half(), 8, cache=cache, dev=DEV)
    gate, up, down = mk(H, F), mk(H, F), mk(F, H)
    norm = FasterTransformerRMSNorm(torch.ones(H).half().to(DEV), 1e-5, cache)
    norm.next_layer = up
    x = torch.randn(M, H, generator=torch.Generator().manual_seed(1)).half().to(DEV)
    for _ in range(3):
        h = norm(x.clone())
        u = up(h, cache)
        g = gate.forward_without_preconditionFusedSilu(h, cache)
        g *= u
        down(g, None, True)
    assert not down.add_outliers
    h = norm(x.clone())
    u = up(h, cache)
    g = gate.forward_without_preconditionFusedSilu(h, cache, amax_for=down)      # maxima of silu(gate) alone ...
    assert getattr(g, "_mixq_row_amax", None) is not None and down._amax_dirty
    g *= u                                                                        # ... but down_proj sees silu(gate) * up
    y = down(g, None, True)
    assert down._amax_dirty, "the stale maxima must not have been consumed"
    h = norm(x.clone()); u = up(h, cache)                                         # (up_proj fills the shared cache again: shape, q_x)
    g2 = gate.forward_without_preconditionFusedSilu(h, cache)
    g2 *= u
    assert torch.equal(y, down(g2, None, True))
    h = norm(x.clone()); u = up(h, cache)
    g3 = gate.forward_without_preconditionFusedSilu(h, cache, mul=u, amax_for=down)   # the next producer run starts from a clean buffer
    y3 = down(g3, None, True)
    assert not down._amax_dirty and int(down._amax_buf.abs().sum()) == 0
    assert (y3.float() - y.float()).abs().max() < 0.05                              # (fused product: one rounding fewer than g *= u)


def test_one_call_plans_are_kept_per_batch_size():
    """A server alternates prefill and decode batches: the frozen layer keeps the argument block of each of the last few batch sizes
    instead of rebuilding it on every switch, and results stay those of the two-call route."""
    layer, cache, cols = frozen_layer(96, 1024, 320, 8, 5, True)
    xs = {M: torch.randn(M, 1024, generator=torch.Generator().manual_seed(M)).half() for M in (96, 16, 40)}
    want = {}
    layer.config.one_call_forward = False
    for M, x in xs.items():
        want[M] = layer(x.clone().to(DEV), None, True).clone()
    layer.config.one_call_forward = True
    plans = {}
    for rnd in range(3):
        for M, x in xs.items():
            y = layer(x.clone().to(DEV), None, True)
            assert torch.equal(y, want[M]), (rnd, M)
            if rnd == 0:
                plans[M] = layer._plan
            else:
                assert layer._plan is plans[M], "the plan of a batch size seen before was rebuilt"
    assert len(layer._plans) == 3
    same = layer.to(DEV)                                              # nothing moves: the kept blocks (and whatever a captured graph addresses) stay
    assert len(same._plans) == 3 and same._plan is not None
    moved = layer.cpu()                                               # a device move drops them (they pin the old tensors)
    assert moved._plans == {} and moved._plan is None


def test_every_wreg_tiling_survives_interleaved_graph_replays():
    """tools/stress_wreg.py: every weights-in-registers tiling (int8 with the wrapped tail, FP6 tuple ring) inside graphs that interleave
    seven shapes, each replay compared bit for bit with the LDS-staged kernels' result: the hazards of a hand-counted register ring
    (a register re-used or copied while its load is in flight) are timing dependent, single launches on quiet buffers can miss them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_wreg.py"), "7", "6"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "TOTAL MISMATCHES 0" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("M,N,K,n_out,act,bias", [(512, 4096, 11008, 110, 0, False), (512, 4096, 14336, 143, 0, True), (500, 4000, 2048, 19, 1, True),
                                                  (130, 200, 128, 0, 0, False), (512, 4096, 4096 + 64, 41, 2, False)])
def test_pairwise_split_k_is_bit_identical(M, N, K, n_out, act, bias):
    """Two workgroups per 128 x 128 tile, half of K each, int32 partial handed over through the workspace (gemm_wreg.hip, KS): integer sums are
    exact in any order, so every output bit equals the one-workgroup-per-tile kernels' - odd k-step counts, ragged M / N, every epilogue term,
    repeated launches (the flags must come back to zero) and graph replay."""
    if "wr128x128_s16_d4_l2_k2" not in _capi.gemm_config_names():
        pytest.skip("the pairwise split-K form lives in the tuning build (make -C mixq_amd/csrc tuning; MIXQ_TUNING_LIB=1): tools/gpu_suite.sh runs it there")
    _capi.ensure_workspace(DEV)
    lib, names = _capi.load(), _capi.gemm_config_names()
    from test_gpu_parity import _fused_case, t
    c = _fused_case(M, N, K, 8, seed=M + N + K, n_out=n_out, bias=bias, addend=act == 2, act=act)
    pad = (n_out + 15) // 16 * 16
    xo = wo = None
    if n_out:
        xo = torch.zeros((M, pad), dtype=torch.float16, device=DEV); xo[:, :n_out] = t(c["xo"]); xo = xo[:, :n_out]
        wo = torch.zeros((N, pad), dtype=torch.float16, device=DEV); wo[:, :n_out] = t(c["wo"]); wo = wo[:, :n_out]
    sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV); sx[:, 0] = t(c["sx"])
    qx, qw = mixlib.PackOperand(t(c["qx"]), 1), mixlib.PackOperand(t(c["qw"]), 2)
    b = None if c["bias"] is None else t(c["bias"])
    ad = None if c["addend"] is None else t(c["addend"])
    sw = t(c["sw"])
    run = lambda out=None: mixlib.FusedLinear(qx, qw, sx, sw, xo, wo, n_out, b, M, N, K, bit=8, act=act, addend=ad, out=out)
    try:
        assert lib.mixq_gemm_set_config(names.index("wr128x128_s16_d4_l2")) == 0
        want = run().clone()
        assert lib.mixq_gemm_set_config(names.index("wr128x128_s16_d4_l2_k2")) == 0
        for rep in range(4):
            got = run()
            torch.cuda.synchronize()
            assert torch.equal(got, want), (rep, int((got != want).sum()))
        out = torch.zeros_like(want)
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for _ in range(3):
                    run(out)
        for _ in range(3):
            out.zero_(); g.replay(); torch.cuda.synchronize()
            assert torch.equal(out, want)
    finally:
        lib.mixq_gemm_set_config(-1)
    ref = O.linear_fused(c["qx"], c["qw"], c["sx"], c["sw"], xo=c["xo"], wo=c["wo"], addend=c["addend"], bias=c["bias"], act=act, bit=8).astype(np.float32)
    assert (np.abs(n(want).astype(np.float32) - ref) <= ulp_tol(ref)).all()


def test_split_k_is_not_chosen_where_it_was_measured_slower_and_refused_when_it_cannot_run():
    if "wr128x128_s16_d4_l2_k2" not in _capi.gemm_config_names():
        pytest.skip("the pairwise split-K form lives in the tuning build (make -C mixq_amd/csrc tuning; MIXQ_TUNING_LIB=1): tools/gpu_suite.sh runs it there")
    lib, names = _capi.load(), _capi.gemm_config_names()
    _capi.ensure_workspace(DEV)
    pick = lambda M, N, K: names[lib.mixq_gemm_pick_config_fmt(M, N, K, 8, 2)]
    # (profiles/r03_splitk_ab.txt: 32.6 vs 28.8 us at 11008 -> 4096: the automatic choice stays with one workgroup per tile)
    for shp in [(512, 4096, 11008), (512, 4096, 14336), (512, 11008, 4096), (512, 4096, 4096), (4096, 4096, 11008), (16, 4096, 11008)]:
        assert "_k2" not in pick(*shp), shp
    # forced onto a problem whose tiles cannot all be resident twice: refused, not deadlocked
    qx = mixlib.PackOperand(torch.zeros((2048, 256), dtype=torch.int8, device=DEV), 1)
    qw = mixlib.PackOperand(torch.zeros((4096, 256), dtype=torch.int8, device=DEV), 2)
    s1 = torch.ones((2048, 1), dtype=torch.float16, device=DEV); s2 = torch.ones((1, 4096), dtype=torch.float16, device=DEV)
    try:
        assert lib.mixq_gemm_set_config(names.index("wr128x128_s16_d4_l2_k2")) == 0
        with pytest.raises(_capi.MixqError):
            mixlib.FusedLinear(qx, qw, s1, s2, None, None, 0, None, 2048, 4096, 256, bit=8)
    finally:
        lib.mixq_gemm_set_config(-1)
