"""Round-4 tests on the MI355X (through the C ABI): what round 3's review asked the host side to hold - kept argument blocks and
hand-over buffers under hipGraph replay, input validation on the one-call route, two threads / two streams on one frozen layer,
a device move after capture - and the kernel-side changes of the round (the panel epilogue's edge cases, the forms that were removed)."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mixq_amd import MixLibCache, MixLinear_GEMM, _capi, mixlib  # noqa: E402
from mixq_amd import linear as L  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_round3 import frozen_layer, n, ulp_tol  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    assert "gfx950" in _capi.device_info()
    _capi.load().mixq_gemm_set_config(-1)
    yield
    _capi.load().mixq_gemm_set_config(-1)


def test_row_maximum_buffer_survives_a_larger_batch_under_graph_replay():
    """gate_proj's GEMM leaves down_proj's row maxima in a buffer down_proj owns.  A graph captured at a small batch has
    that buffer's address baked in; a later, larger eager batch used to re-allocate it, and the replay then wrote into - and zeroed -
    freed memory.  The buffer is now allocated once at the cache's row capacity: replay after the larger batch is still exact."""
    from mixq_amd import FasterTransformerRMSNorm, MixLlamaMLP, fused
    H, F, CAP = 512, 1536, 96
    torch.manual_seed(0)
    cache = MixLibCache(CAP, device=DEV)
    mk = lambda k, nn_: MixLinear_GEMM.from_linear(torch.nn.Linear(k, nn_, bias=False).half(), 8, cache=cache, dev=DEV)
    gate, up, down = mk(H, F), mk(H, F), mk(F, H)
    inner = MixLlamaMLP(gate, down, up, cache)
    norm = FasterTransformerRMSNorm((torch.rand(H) + 0.5).half().to(DEV), 1e-5, cache)
    norm.next_layer = up
    mlp = lambda x: inner(norm(x))
    g = torch.Generator().manual_seed(1)
    cols = torch.randperm(H, generator=g)[:5]

    def batch(M):
        x = torch.randn(M, H, generator=g).half()
        x[:, cols] *= 20
        return x.to(DEV)
    prev = inner.config.fuse_down_amax
    try:
        inner.config.fuse_down_amax = True
        for _ in range(3):
            mlp(batch(CAP))                                            # freeze the outlier search of every layer
        assert not down.add_outliers
        xs = batch(16)
        y_small = mlp(xs.clone())
        buf = down._amax_buf
        assert buf is not None and buf.numel() >= CAP, "the hand-over buffer must be sized for the cache's capacity from the start"
        side = torch.cuda.Stream()
        xg = xs.clone()
        with torch.cuda.stream(side):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                yg = mlp(xg)
        torch.cuda.synchronize()
        y_big = mlp(batch(CAP))                                        # a larger eager batch between capture and replay
        assert down._amax_buf is buf and down._amax_buf.data_ptr() == buf.data_ptr(), "the buffer a captured graph addresses was replaced"
        junk = [torch.full((CAP,), 0x7fffffff, dtype=torch.int32, device=DEV) for _ in range(8)]   # whatever the allocator hands out next
        for _ in range(3):
            xg.copy_(xs)
            with torch.cuda.stream(side):
                gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(yg, y_small)
        assert int(buf.abs().sum()) == 0 and all(int(j[0]) == 0x7fffffff for j in junk)
        del y_big
    finally:
        inner.config.fuse_down_amax = prev


def test_one_call_route_validates_its_input_on_every_call():
    """the kept argument block takes x by address.  A frozen layer must refuse what the two-call route's QuantFused refused:
    another dtype, another column count, a strided last dimension, a host tensor - also when a plan for the same M and row stride exists."""
    M, K, N = 64, 512, 256
    layer, cache, cols = frozen_layer(M, K, N, 8, 5, False)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(3)).half().to(DEV)
    y = layer(x.clone(), None, True)
    assert layer._plan is not None
    with pytest.raises(RuntimeError, match="float16"):
        layer(x.float().half().bfloat16(), None, True)                  # same M, same row stride in elements: the cached plan's key
    with pytest.raises(RuntimeError, match="float16"):
        layer(x.float()[:, :K], None, True)
    with pytest.raises(RuntimeError, match="columns"):
        layer(torch.randn(M, K + 64, device=DEV).half(), None, True)
    wide = torch.randn(M, 2 * K, device=DEV).half()
    with pytest.raises(RuntimeError, match="contiguous last dimension"):
        layer(wide[:, ::2], None, True)
    with pytest.raises(RuntimeError, match="GPU"):
        layer(x.cpu(), None, True)
    assert torch.equal(layer(x.clone(), None, True), y)                # ... and the layer still works
    # the plan itself refuses a tensor it was not built for (another row count / stride), whoever calls it
    with pytest.raises(RuntimeError, match="ForwardPlan"):
        layer._plan.run(x[: M // 2])
    with pytest.raises(RuntimeError, match="ForwardPlan"):
        layer._plan.run(wide[:, :K])


def test_two_threads_two_streams_on_one_frozen_layer():
    """ADVICE r03 / VERDICT r03: ForwardPlan.run fills ONE argument block and makes a foreign call that releases the GIL.  Two threads,
    each on its own stream, running the same frozen layer (same cache, same plan) must not launch with each other's pointers."""
    M, K, N = 64, 1024, 512
    layer, cache, cols = frozen_layer(M, K, N, 8, 7, True)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(5)).half()
    x[:, cols] *= 20
    x = x.to(DEV)
    y_ref = layer(x.clone(), None, True)
    plan = layer._plan
    errors = []

    def worker(seed):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(150):
                    y = layer(x.clone(), None, True)                   # (a fresh copy: the forward zeroes the outlier columns of its input in place)
                    if it % 10 == 0:
                        st.synchronize()
                        if not torch.equal(y, y_ref):
                            errors.append((seed, it))
                st.synchronize()
        except Exception as e:                                         # noqa: BLE001
            errors.append((seed, repr(e)))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    assert layer._plan is plan, "the two threads must have shared the one kept argument block"


def test_two_layers_sharing_a_cache_issued_from_two_streams():
    """two layers that share one MixLibCache (its x_scale rows, its outlier scratch), each issued from its own
    stream with the hand-over ordered by an event as a model with a side stream would: same bits as on one stream."""
    M, K = 48, 512
    torch.manual_seed(0)
    cache = MixLibCache(M, device=DEV)
    a = MixLinear_GEMM.from_linear(torch.nn.Linear(K, 384, bias=False).half(), 8, cache=cache, dev=DEV)
    b = MixLinear_GEMM.from_linear(torch.nn.Linear(384, 256, bias=True).half(), 8, cache=cache, dev=DEV)
    g = torch.Generator().manual_seed(2)
    cols = torch.randperm(K, generator=g)[:4]
    xs = []
    for _ in range(4):
        x = torch.randn(M, K, generator=g).half()
        x[:, cols] *= 20
        xs.append(x.to(DEV))
    for x in xs[:3]:
        b(a(x.clone(), None, True), None, True)
    assert not a.add_outliers and not b.add_outliers
    y_ref = b(a(xs[3].clone(), None, True), None, True)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        xi = xs[3].clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            h = a(xi, None, True)
            ev = torch.cuda.Event()
            ev.record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            h.record_stream(s2)
            y = b(h, None, True)
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref)


def test_moving_a_captured_layer_raises_instead_of_leaving_a_graph_with_dangling_addresses():
    """a hipGraph captured through a frozen layer replays the addresses of its weight image, outlier operands and
    device count.  .to(another device) frees them; the graph cannot be stopped from replaying - so the MOVE is what raises, until the
    caller says the graph is gone.  A layer that never ran under capture moves freely, and rebuilds its plans where it lands."""
    M, K, N = 32, 512, 256
    layer, cache, cols = frozen_layer(M, K, N, 8, 3, False)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(9)).half().to(DEV)
    y_ref = layer(x.clone(), None, True)
    side = torch.cuda.Stream()
    xg = x.clone()
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            yg = layer(xg, None, True)
    torch.cuda.synchronize()
    assert layer._plan.captured
    with pytest.raises(RuntimeError, match="hipGraph capture"):
        layer.cpu()
    with torch.cuda.stream(side):
        gr.replay()                                                    # the refused move left everything in place: the graph is still good
    torch.cuda.synchronize()
    assert torch.equal(yg, y_ref)
    layer.half()                                                       # (an _apply that moves nothing is not a move)
    del gr
    layer.allow_move_after_capture = True
    moved = layer.cpu()
    assert moved._plans == {} and moved._plan is None and not moved._wpk.is_cuda
    back = moved.to(DEV)
    assert torch.equal(back(x.clone(), None, True), y_ref)
    free, _, _ = frozen_layer(M, K, N, 8, 3, False, seed=4)            # never captured: no ceremony
    free.cpu()


def test_the_128x256_tile_has_no_nibble_form():
    """Round 4 removed gemm_wreg_kernel<8, 4, ., ., 1, ...>: 128 accumulators + the weight ring + the expanded nibble fragments never
    fitted 256 registers (scratch inside the hand-counted region).  A forced 128 x 256 configuration answers MIXQ_EINVAL for nibble
    operands - not a silent fallback - and the automatic choice never lands on it for them."""
    lib = _capi.load()
    names = _capi.gemm_config_names()
    M, N, K = 64, 512, 512
    g = torch.Generator().manual_seed(0)
    qx = torch.randint(0, 256, (M, K // 2), generator=g, dtype=torch.uint8).to(DEV)
    qw = torch.randint(0, 256, (N, K // 2), generator=g, dtype=torch.uint8).to(DEV)
    sx = torch.full((M, 1), 2.0 ** -3, dtype=torch.float16, device=DEV)
    sw = torch.full((1, N), 2.0 ** -3, dtype=torch.float16, device=DEV)
    xp, wp = mixlib.PackOperand(qx, 1), mixlib.PackOperand(qw, 2)
    try:
        for nm in ("wr128x256_s16_d3_l2", "wr128x256_s8_d3_l1"):
            assert lib.mixq_gemm_set_config(names.index(nm)) == 0
            with pytest.raises(_capi.MixqError):
                mixlib.FusedLinear(xp, wp, sx, sw, None, None, 0, None, M, N, K, bit=4)
    finally:
        lib.mixq_gemm_set_config(-1)
    for (m, nn_, k) in [(512, 14336, 4096), (512, 28672, 8192), (2048, 11008, 4096)]:
        assert "128x256" not in names[lib.mixq_gemm_pick_config_fmt(m, nn_, k, 4, 2)]


@pytest.mark.parametrize("M,N,K,n_out,bias,act", [(512, 11008, 4096, 41, False, 0), (200, 584, 512, 70, True, 1), (130, 204, 256, 33, True, 0),
                                                   (96, 192, 128, 0, False, 2), (33, 1000, 1024, 129, False, 0)])
def test_panel_epilogue_edge_cases_against_the_oracle(M, N, K, n_out, bias, act):
    """The panel-pipelined epilogue of gemm_wreg.hip on what its fast path does not see in the other suites together: ragged last
    panels (M % 32 != 0), tiles hanging over N, outputs that cannot be staged (N % 8 != 0: 8-byte stores from the consumers), more
    outlier columns than went through LDS (> 64: the extra k-steps from global memory), no outlier columns at all, and the optional
    terms - against the oracle at <= 2 ulp, every weights-in-registers tiling."""
    from test_gpu_parity import _fused_case, _run_fused, _wr_configs
    lib = _capi.load()
    names = _capi.gemm_config_names()
    c = _fused_case(M, N, K, 8, seed=M + N, n_out=n_out, bias=bias, addend=act == 2, act=act)
    ref = O.linear_fused(c["qx"], c["qw"], c["sx"], c["sw"], xo=c["xo"], wo=c["wo"], addend=c["addend"], bias=c["bias"], act=act,
                         bit=8).astype(np.float32)
    try:
        for cfg in _wr_configs():
            if M * N > 2e6 and not names[cfg].startswith("wr128x192_s16_d4"):
                continue                                               # (the full-size case on the metric tiling only)
            assert lib.mixq_gemm_set_config(cfg) == 0
            y = n(_run_fused(c, 2)).astype(np.float32)
            assert np.isfinite(y).all(), names[cfg]
            assert (np.abs(y - ref) <= ulp_tol(ref)).all(), (names[cfg], float(np.abs(y - ref).max()))
    finally:
        lib.mixq_gemm_set_config(-1)


@pytest.mark.parametrize("bit,fmt,M,K,ncols,cap", [(8, 1, 512, 4096, 41, 48), (8, 0, 37, 512, 3, 3), (8, 1, 100, 11008, 130, 144), (4, 4, 96, 1024, 128, 128),
                                                   (4, 1, 64, 512, 17, 32), (8, 1, 16, 256, 1, 16)])
def test_quantise_pass_with_the_kept_column_mask_writes_the_same_bytes(bit, fmt, M, K, ncols, cap):
    """mixq_quant_fused_masked (the frozen layer's route: its bit-per-column mask of `ind` is kept in device memory) against
    mixq_quant_fused, which builds that mask in LDS on every launch: q_x, x_scale, x_out, the zeroed columns of x and the misprediction
    flag, with the live count in device memory below the capacity of `ind` (poison behind it)."""
    g = torch.Generator().manual_seed(bit * 1000 + K + ncols)
    x = torch.randn(M, K, generator=g).half()
    cols = torch.randperm(K, generator=g)[:ncols].to(torch.int32)      # (unsorted, as `ind` is after an append: linear.py:214)
    x[:, cols.long()] *= 30
    ind = torch.full((cap,), K - 1, dtype=torch.int32)                   # (entries behind the live count: never read as columns)
    ind[:ncols] = cols
    ind, n_dev = ind.to(DEV), torch.tensor([ncols], dtype=torch.int32, device=DEV)
    mask = L.kept_outlier_map(ind[:ncols], K)                            # bits, count word, per-column AND-masks (include/mixq_hip.h)
    outs = []
    for cm in (None, mask):
        xd = x.clone().to(DEV)
        sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        q, xo = mixlib.QuantFused(xd, ind, sx, bit, 6.0, flag=flag, n_dev=n_dev if cap > ncols else None, fmt=fmt, col_mask=cm)
        torch.cuda.synchronize()
        if fmt:                                                         # (the packed image's pad rows beyond M are not written: compare the rows)
            q = mixlib.UnpackOperand(q, M)
        outs.append((q.clone(), sx, xo[:, :ncols].clone(), xd, flag))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert int((outs[1][3][:, cols.long().to(DEV)] != 0).sum()) == 0


def test_frozen_layer_hands_its_kept_mask_to_the_one_call_forward():
    """The plan of a frozen layer carries the layer's column mask; the forward equals the two-call route's bit for bit."""
    M, K, N = 96, 1024, 512
    layer, cache, cols = frozen_layer(M, K, N, 8, 7, True, seed=11)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(5)).half()
    x[:, cols] *= 20
    y1 = layer(x.clone().to(DEV), None, True)
    assert layer._plan is not None and layer._plan.kept_mask is not None and layer._plan.kept_mask is layer._col_mask()
    layer.config.one_call_forward = False
    try:
        y2 = layer(x.clone().to(DEV), None, True)
    finally:
        layer.config.one_call_forward = True
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("bit,fmt,M,K,ncols,cap", [(8, 1, 512, 4096, 41, 48), (8, 0, 37, 512, 3, 3), (4, 4, 96, 1024, 128, 128), (8, 1, 40, 8192, 300, 304)])
def test_fused_norm_with_the_kept_column_mask_writes_the_same_bytes(bit, fmt, M, K, ncols, cap):
    """mixq_rmsnorm_quant_fused_masked against mixq_rmsnorm_quant_fused: normalised output, q_x, x_scale, x_out - also with more outlier
    columns than the workgroup has threads' first entries (300) and the live count below the capacity of `ind`."""
    g = torch.Generator().manual_seed(bit * 1000 + K + ncols)
    x = torch.randn(M, K, generator=g).half()
    cols = torch.randperm(K, generator=g)[:ncols].to(torch.int32)      # (unsorted, as `ind` is after an append: linear.py:214)
    x[:, cols.long()] *= 30
    wgt = (torch.rand(K, generator=g) + 0.5).half().to(DEV)
    ind = torch.full((cap,), K - 1, dtype=torch.int32)
    ind[:ncols] = cols
    ind, n_dev = ind.to(DEV), torch.tensor([ncols], dtype=torch.int32, device=DEV)
    mask = L.kept_outlier_map(ind[:ncols], K)                            # bits, count word, per-column AND-masks (include/mixq_hip.h)
    outs = []
    for cm in (None, mask):
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
