"""Round 6 GPU tests: the G8 fixtures (the reference's own norm.py + mlp.py + linear.py flow, oracle/gen_golden_g8.py) on the HIP backend
- through the operator mirror step by step, through MixLlamaMLP.forward with the joint gate / up launch on and off, and native by native
through the `mixlib` shim with the operands the reference's flow handed each native."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import g8_replay  # noqa: E402
from mixq_amd import MixLibCache, MixqConfig, _capi, mixlib  # noqa: E402
from mixq_amd.mixlib import fmt_of  # noqa: E402
from test_pack_properties import packed_unpack  # noqa: E402

DEV = "cuda"
G8 = ["g8_mlp_block_w8.npz", "g8_mlp_block_w4.npz"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def n(x):
    return x.detach().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    assert "gfx950" in _capi.device_info()
    _capi.load().mixq_gemm_set_config(-1)
    yield
    _capi.load().mixq_gemm_set_config(-1)


def _unpacked_q(cache, M, KB):
    q = n(cache.q_xcache)
    fmt = fmt_of(cache.q_xcache)
    if fmt:
        return packed_unpack(q.reshape(-1).view(np.uint8), M, KB, fmt)
    return q.view(np.uint8)[:M]


@pytest.mark.parametrize("name", G8)
def test_g8_block_trace_step_by_step_on_gpu(golden, name):
    """Every step of mlp.py:57-70 behind the fused norm on the HIP kernels: the state the reference's flow left (ind, weight_cache, cnt,
    forward_without_precondition_len, cache.new_ind, x_scale, q_xcache, activation_outliers, the in-place zeroed activation) bit-exact."""
    worst = g8_replay.replay_walk(golden(name), DEV, _unpacked_q)
    assert worst <= 4e-3


@pytest.mark.parametrize("joint", [True, False])
@pytest.mark.parametrize("name", G8)
def test_g8_block_trace_through_forward_on_gpu(golden, name, joint):
    """MixLlamaMLP.forward as the product runs it - gate_proj + up_proj as ONE launch once both predictions are frozen (calls 2, 3 of the
    8-bit trace; every call of the 4-bit one) or as two launches - against the reference's y (<= 1e-2) and final layer state (bit-exact)."""
    worst = g8_replay.replay_forward(golden(name), DEV, config=MixqConfig(joint_gate_up=joint))
    assert worst <= 1e-2


def test_g8_natives_through_the_mixlib_shim(golden):
    """Option A: the reference's Python calling `mixlib`.  /root/reference does not exist on the GPU box, so the
    reference's flow is replayed NATIVE BY NATIVE from the 8-bit fixture: each shim entry point gets the operands the reference's own flow
    handed that native at that point (recorded `ind`, `weight_cache`, the call order in c{i}_calls) and must return what the flow went on
    with - integer / byte results bit-exact, fp16 outputs <= 4e-3 (two roundings against one)."""
    g = golden("g8_mlp_block_w8.npz")
    I, K = g["up_weight"].shape
    x_scale = torch.zeros((64, 1), dtype=torch.float16, device=DEV)
    zeros = MixLibCache(64, device=DEV).zeros                                      # (cache.zeros as the addend, linear.py:241,272: recognised by its tag)
    norm_w = t(g["norm_weight"])
    qw = {k: t(g[k + "_q_weight"]) for k in ("up", "gate", "down")}
    sw = {k: t(g[k + "_scale_col"]) for k in ("up", "gate", "down")}
    prev_ind = np.zeros((0,), np.int32)
    for i in range(int(g["ncalls"])):
        calls = [str(c) for c in g[f"c{i}_calls"]]
        x = t(g[f"c{i}_x_in"].copy())
        M = x.numel() // K
        out = torch.empty_like(x)
        assert calls[0] == "layernorm_forward_cuda_extract_outliers"
        ao, q = mixlib.layernorm_forward_cuda_extract_outliers(x, norm_w, out, float(g["eps"]), t(prev_ind), x_scale)   # norm.py:25-28
        assert np.array_equal(bits(n(out)), bits(g[f"c{i}_n_hidden"])) and np.array_equal(bits(n(x_scale)[:M]), bits(g[f"c{i}_n_x_scale"]))
        assert np.array_equal(n(q)[:M].view(np.uint8), g[f"c{i}_n_q_xcache"].view(np.uint8))
        if prev_ind.size:
            assert np.array_equal(bits(n(ao)), bits(g[f"c{i}_n_activation_outliers"]))
        inputs = out.reshape(-1, K)
        if "ExtractOutliersAndSetToZeros" in calls:                               # linear.py:203-221: new columns found by up_proj_
            new_ind = t(g[f"c{i}_new_ind"])
            new = mixlib.ExtractOutliersAndSetToZeros(new_ind, inputs)
            wc_new = mixlib.DequantWeightCols(qw["up"], sw["up"], new_ind, 8)       # (q_weight[:, ind].half() * scale_col.T, linear.py:207)
            assert np.array_equal(bits(n(wc_new)), bits(g[f"c{i}_up_weight_cache"][:, prev_ind.size:]))
            ao = torch.hstack((ao, new)) if prev_ind.size else new
            q = mixlib.FindRowScale(inputs, x_scale, M, K, 8)
            assert np.array_equal(bits(n(inputs)), bits(g[f"c{i}_u_hidden_after"].reshape(-1, K)))
        ind_now = g[f"c{i}_up_ind"]
        assert np.array_equal(bits(n(x_scale)[:M]), bits(g[f"c{i}_u_x_scale"])) and np.array_equal(n(q)[:M].view(np.uint8), g[f"c{i}_u_q_xcache"].view(np.uint8))
        if ind_now.size:
            assert np.array_equal(bits(n(ao)), bits(g[f"c{i}_u_activation_outliers"]))
            add_u = torch.mm(ao, t(g[f"c{i}_up_weight_cache"]).T)                 # linear.py:248
            add_g = torch.mm(ao, t(g[f"c{i}_gate_weight_cache"]).T)
        else:
            add_u = add_g = zeros
        y_up = mixlib.int8FusedDequantize(q, qw["up"], x_scale, sw["up"], add_u, M, I, K)
        y_g = mixlib.int8FusedDequantizeSilu(q, qw["gate"], x_scale, sw["gate"], add_g, M, I, K)
        assert np.abs(n(y_up).astype(np.float32) - g[f"c{i}_u_y"].reshape(M, I).astype(np.float32)).max() <= 4e-3
        assert np.abs(n(y_g).astype(np.float32) - g[f"c{i}_g_y"].reshape(M, I).astype(np.float32)).max() <= 4e-3
        prod = t(g[f"c{i}_prod"]).reshape(M, I).clone()                           # down_proj_ from the reference's own product
        qd = mixlib.FindRowScale(prod, x_scale, M, I, 8)                           # linear.py:190 (down_proj_ found no outlier column)
        assert np.array_equal(bits(n(x_scale)[:M]), bits(g[f"c{i}_d_x_scale"])) and np.array_equal(n(qd)[:M].view(np.uint8), g[f"c{i}_d_q_xcache"].view(np.uint8))
        y = mixlib.int8FusedDequantize(qd, qw["down"], x_scale, sw["down"], zeros, M, K, I)
        assert np.abs(n(y).astype(np.float32) - g[f"c{i}_y"].reshape(M, K).astype(np.float32)).max() <= 4e-3
        assert g[f"c{i}_down_ind"].size == 0
        prev_ind = ind_now


@pytest.mark.parametrize("bit", [8, 4])
def test_operator_state_sequence_on_gpu(bit):
    """The GPU twin of tests/test_operator_stateful.py: one fixed sequence (search, a new outlier column, freeze, joint gate / up route, a
    layer called on its own, state_dict round trip, deepcopy, the joint route off and on, new weights, .half(), a 3-row batch) on the HIP
    backend - the block with every caching feature on returns the bytes of the twin that keeps nothing between forwards, step by step."""
    import test_operator_stateful as S

    class Gpu(S.BlockMachine if bit == 8 else S.BlockMachine4):
        DEV, BACKEND = DEV, None
    S.fixed_sequence(Gpu())


def test_falcon_and_gptj_mlp_wrappers_on_gpu():
    """mixquant/modules/fused/mlp.py:8-32, :75-93 on the HIP backend: the wrappers equal the operator calls written out by hand bit for bit, and the
    dense fp32 block within the W8A8 quantisation error (the parity gate of the Linears themselves is tests/test_gpu_parity.py's)."""
    from mixq_amd import FasterTransformerRMSNorm, MixFalconMLP, MixGPTJMLP, MixLinear_GEMM
    from transformers.activations import ACT2FN
    torch.manual_seed(11)
    K, I, M = 512, 2048, 96
    a, b = torch.nn.Linear(K, I).half(), torch.nn.Linear(I, K).half()
    cache = MixLibCache(128, device=DEV)
    mk = lambda l, **kw: MixLinear_GEMM.from_linear(l, 8, cache=cache, dev=DEV, **kw)
    x = torch.randn(M, K).half().to(DEV)
    x[:, [5, 77]] *= 30

    def normed(first):
        norm = FasterTransformerRMSNorm(torch.ones(K, dtype=torch.float16, device=DEV), cache=cache)
        norm.next_layer = first
        return norm(x.clone())

    for _ in range(3):                                      # call 0 finds the outlier columns, later calls run the frozen (one-call) forwards
        l1, l2 = (mk(a), mk(b)) if _ == 0 else (l1, l2)
        y = MixFalconMLP(l1, l2, cache)(normed(l1))
    r1, r2 = mk(a), mk(b)
    for _ in range(3):
        ref = r2(torch.nn.GELU()(r1(normed(r1), cache)), cache, True)
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    xf = x.float().cpu()
    xn = xf / torch.sqrt((xf ** 2).mean(dim=-1, keepdim=True) + 1e-6)
    dense = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xn, a.weight.float(), a.bias.float())), b.weight.float(), b.bias.float())
    assert (y.float().cpu() - dense).abs().max() < 0.05 * dense.abs().max() + 0.05

    class Cfg:
        activation_function, resid_pdrop = "gelu_new", 0.0

    class Mod(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc_in, self.fc_out = mk(a), mk(b, weight_only=True, name="fc_out")

    m, m2 = Mod(), Mod()
    g = MixGPTJMLP(m, Cfg(), cache).eval()
    for _ in range(3):
        y2 = g(normed(m.fc_in))
        ref2 = m2.fc_out(ACT2FN["gelu_new"](m2.fc_in(normed(m2.fc_in), cache)), cache)
    torch.cuda.synchronize()
    assert torch.equal(y2, ref2)
    dense2 = torch.nn.functional.linear(ACT2FN["gelu_new"](torch.nn.functional.linear(xn, a.weight.float(), a.bias.float())), b.weight.float(), b.bias.float())
    assert (y2.float().cpu() - dense2).abs().max() < 0.05 * dense2.abs().max() + 0.05


@pytest.mark.parametrize("bit", [8, 4])
def test_outlier_count_edges_through_the_counted_drain(bit):
    """Round 6: the loader waves request the epilogue's scales and ALL TQ tail k-steps of X_out during the drain of the ring, with compile-time
    request counts (gemm_wreg.hip, `CDRAIN`): chunks past the live columns read column 0 and are zeroed by the LDS fix-up, both loader waves load the
    scales, the bias slot is always loaded.  Every outlier count around the 16- / 32-column boundaries, with the count in the argument and in device memory
    (capacity larger than the count: poison beyond it), K long enough for the counted path, ragged M / N - every weights-in-registers tiling against the
    oracle and bit-identical to each other."""
    from test_gpu_parity import _fused_case, _run_fused, _wr_configs, ulp_tol
    from oracle import oracle as O
    from mixq_amd._capi import FMT_F6X128, FMT_R6X128
    lib, names = _capi.load(), _capi.gemm_config_names()
    M, N, K = 130, 328, 2048                                 # 32 int8 k-steps (ring 14 deep), 16 FP6 k-steps (ring 6 deep)
    try:
        for n_out in ([1, 15, 16, 17, 31, 32, 33, 47, 64, 65, 96, 128, 143] if bit == 8 else [16, 33, 64, 96, 128, 143]):
            for bias, cap in ((False, 0), (True, 0), (False, ((n_out + 15) // 16 * 16) + 16)):
                c = _fused_case(M, N, K, bit, seed=7 * n_out + bit, n_out=n_out, bias=bias, addend=False, act=0)
                ref = O.linear_fused(c["qx"], c["qw"], c["sx"], c["sw"], xo=c["xo"], wo=c["wo"], bias=c["bias"], bit=bit).astype(np.float32)
                first = None
                cfgs = _wr_configs() if bit == 8 else [names.index(nm) for nm in ("wr128x192_s16_d4_l2", "wr128x128_s16_d4_l2", "wr64x128_s16_d4_l2", "wr64x192_s16_d4_l2", "wr64x256_s16_d4_l2", "wr32x64_s8_d6_l1")]
                for cfg in cfgs:
                    assert lib.mixq_gemm_set_config(cfg) == 0
                    if bit == 8:
                        y = _run_fused(c, 2, n_dev_cap=cap)
                    else:                                    # W4A4 on the FP6 pipe: both operands as FP6 codes
                        pad = (max(n_out, cap) + 15) // 16 * 16
                        xo = torch.full((M, pad), float("nan"), dtype=torch.float16, device=DEV); xo[:, :n_out] = t(c["xo"])
                        wo = torch.full((N, pad), float("nan"), dtype=torch.float16, device=DEV); wo[:, :n_out] = t(c["wo"])
                        ncap = max(n_out, cap)
                        n_dev = torch.tensor([n_out], dtype=torch.int32, device=DEV) if cap else None
                        sx = torch.zeros((M, 1), dtype=torch.float16, device=DEV); sx[:, 0] = t(c["sx"])
                        y = mixlib.FusedLinear(mixlib.PackOperand(t(c["qx"]), FMT_R6X128), mixlib.PackOperand(t(c["qw"]), FMT_F6X128), sx, t(c["sw"]), xo[:, :ncap], wo[:, :ncap], ncap,
                                               None if c["bias"] is None else t(c["bias"]), M, N, K, bit=4, n_out_dev=n_dev)
                    torch.cuda.synchronize()
                    yn = n(y).astype(np.float32)
                    assert np.isfinite(yn).all(), (names[cfg], n_out, cap)
                    assert (np.abs(yn - ref) <= ulp_tol(ref)).all(), (names[cfg], n_out, cap, float(np.abs(yn - ref).max()))
                    if first is None:
                        first = y.clone()
                    else:
                        assert torch.equal(y, first), (names[cfg], n_out, cap)
    finally:
        lib.mixq_gemm_set_config(-1)
