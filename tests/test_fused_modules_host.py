"""Host logic of the modules either side of the path on the oracle backend: the fused
RMSNorm hands the next linear exactly what its own unfused pre-pass would have produced, and the MLP wrapper calls the
operator the way mixquant/modules/fused/mlp.py:57-70 does."""
import numpy as np
import pytest
from conftest import swap_backend
import torch

import backend_oracle
import mixq_amd.fused as F
import mixq_amd.linear as L
from mixq_amd import FasterTransformerRMSNorm, MixFalconMLP, MixGPTJMLP, MixLibCache, MixLinear_GEMM, MixLlamaMLP
from oracle import oracle as O


@pytest.fixture()
def oracle_backend():
    p1, p2 = swap_backend(L, backend_oracle), swap_backend(F, backend_oracle)
    backend_oracle.calls.clear()
    yield backend_oracle
    swap_backend(L, p1)
    swap_backend(F, p2)


def test_rmsnorm_oracle_matches_fp64_definition():
    rng = np.random.default_rng(0)
    for M, K in [(3, 64), (5, 4096), (2, 11008)]:
        x = (rng.standard_normal((M, K)) * 3).astype(np.float16)
        w = (1 + 0.2 * rng.standard_normal(K)).astype(np.float16)
        y = O.rmsnorm(x, w, 1e-6).astype(np.float64)
        xd = x.astype(np.float64)
        ref = xd / np.sqrt((xd ** 2).mean(axis=1, keepdims=True) + 1e-6) * w.astype(np.float64)
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10)
        assert (np.abs(y - ref) <= 0.51 * ulp + 1e-7 * np.abs(ref)).all()      # correctly rounded up to the fp32 path's noise


def test_fused_norm_feeds_the_next_linear(oracle_backend):
    torch.manual_seed(0)
    K, N, M = 256, 96, 24
    lin = torch.nn.Linear(K, N, bias=False).half()
    cache = MixLibCache(64, device="cpu")
    wpack = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev="cpu")
    norm = FasterTransformerRMSNorm(torch.ones(K) + 0.1 * torch.randn(K), eps=1e-6, cache=cache)
    norm.next_layer = wpack
    ref_layer = MixLinear_GEMM.from_linear(lin, 8, cache=MixLibCache(64, device="cpu"), dev="cpu")
    plain = FasterTransformerRMSNorm(norm.weight.clone(), eps=1e-6, cache=None)
    cols = [7, 100, 201]
    for call in range(3):
        h = torch.randn(2, M // 2, K, generator=torch.Generator().manual_seed(call)).half()
        h[..., cols] *= 25
        h0 = h.clone()
        hidden = norm(h)                                 # fills cache.q_xcache / x_scale / activation_outliers
        assert torch.equal(h, h0), "the norm must not modify its input"
        y = wpack(hidden, None, False)                   # unfused=False: starts from the cache (attn.py:219)
        # reference flow: plain norm, then the linear's own unfused pre-pass
        hidden_ref = plain(h0.clone())
        y_ref = ref_layer(hidden_ref, None, True)
        assert torch.equal(y, y_ref)
        assert torch.equal(wpack.ind, ref_layer.ind) and wpack.ind.tolist() == cols
        assert torch.equal(hidden, hidden_ref), "both flows leave the normalised activation with the outlier columns zeroed"
    assert oracle_backend.calls.count("RMSNormQuantFused") == 3


def test_mlp_wrapper_matches_composition(oracle_backend):
    torch.manual_seed(1)
    K, I, M = 128, 256, 8
    up, gate, down = (torch.nn.Linear(K, I, bias=False).half(), torch.nn.Linear(K, I, bias=False).half(),
                      torch.nn.Linear(I, K, bias=False).half())
    cache = MixLibCache(16, device="cpu")
    mk = lambda l: MixLinear_GEMM.from_linear(l, 8, cache=cache, dev="cpu")
    up_q, gate_q, down_q = mk(up), mk(gate), mk(down)
    norm = FasterTransformerRMSNorm(torch.ones(K), cache=cache)
    norm.next_layer = up_q
    mlp = MixLlamaMLP(gate_q, down_q, up_q, cache)
    x = torch.randn(M, K).half()
    y = mlp(norm(x))
    assert tuple(y.shape) == (M, K)
    ref = torch.nn.functional.linear(
        torch.nn.functional.silu(torch.nn.functional.linear(norm_ref(x), gate.weight.float())) *
        torch.nn.functional.linear(norm_ref(x), up.weight.float()), down.weight.float())
    assert (y.float() - ref).abs().max() < 0.05 * ref.abs().max() + 0.05     # W8A8 quantisation error scale, not the parity gate


def norm_ref(x):
    xf = x.float()
    return xf / torch.sqrt((xf ** 2).mean(dim=-1, keepdim=True) + 1e-6)


def test_pair_row_order_of_the_joint_gate_up_launch():
    """mixq_amd.fused.interleave_pair_rows (what MixLlamaMLP builds its joint image with) against the oracle's element-by-element
    definition of the MIXQ_ACT_SILU_PAIR row order, and the oracle's paired compute step against the two steps it stands for."""
    rng = np.random.default_rng(3)
    for shape in [(8,), (16, 5), (96, 32)]:
        up = rng.integers(-100, 100, size=shape).astype(np.int8)
        gate = rng.integers(-100, 100, size=shape).astype(np.int8)
        j = F.interleave_pair_rows(torch.from_numpy(up), torch.from_numpy(gate)).numpy()
        assert np.array_equal(j, O.pair_rows_interleave(up, gate))
        u2, g2 = F.split_pair_rows(torch.from_numpy(j))
        assert np.array_equal(u2.numpy(), up) and np.array_equal(g2.numpy(), gate)
        u3, g3 = O.pair_rows_split(j)
        assert np.array_equal(u3, up) and np.array_equal(g3, gate)
    with pytest.raises(ValueError):
        F.interleave_pair_rows(torch.zeros(3, 2), torch.zeros(3, 2))
    M, N, K, n_out = 10, 24, 64, 5
    qx = rng.integers(-127, 128, size=(M, K)).astype(np.int8)
    sx = (rng.random(M) * 0.01 + 0.001).astype(np.float16)
    xo = rng.standard_normal((M, n_out)).astype(np.float16)
    lay = []
    for _ in range(2):
        lay.append(dict(qw=rng.integers(-127, 128, size=(N, K)).astype(np.int8), sw=(rng.random(N) * 0.01 + 0.001).astype(np.float16),
                        wo=(rng.standard_normal((N, n_out)) * 0.1).astype(np.float16), b=rng.standard_normal(N).astype(np.float16)))
    u, g = lay
    up = O.linear_fused(qx, u["qw"], sx, u["sw"], xo=xo, wo=u["wo"], bias=u["b"], act=0)
    ref = O.linear_fused(qx, g["qw"], sx, g["sw"], xo=xo, wo=g["wo"], addend=up, bias=g["b"], act=2)
    got = O.linear_fused_pair(qx, O.pair_rows_interleave(u["qw"], g["qw"]), sx, O.pair_rows_interleave(u["sw"], g["sw"]), xo=xo,
                              wo2=O.pair_rows_interleave(u["wo"], g["wo"]), bias2=O.pair_rows_interleave(u["b"], g["b"]))
    assert got.shape == (M, N) and np.array_equal(got.view(np.uint16), ref.view(np.uint16))
    # ... which is silu(gate) * up of the fp64 definition within fp16 resolution
    z = lambda d: (qx.astype(np.float64) @ d["qw"].astype(np.float64).T) * sx.astype(np.float64)[:, None] * d["sw"].astype(np.float64)[None, :] \
        + xo.astype(np.float64) @ d["wo"].astype(np.float64).T
    zu, zg = z(u) + u["b"].astype(np.float64), z(g)
    want = (zg / (1 + np.exp(-zg)) + g["b"].astype(np.float64)) * zu
    assert np.abs(got.astype(np.float64) - want).max() <= 4e-3 * np.abs(want).max() + 1e-3


# ---- G8: the reference's own norm.py + mlp.py + linear.py flow, recorded by oracle/gen_golden_g8.py -------------------------------
@pytest.mark.parametrize("name", ["g8_mlp_block_w8.npz", "g8_mlp_block_w4.npz"])
def test_g8_block_trace_step_by_step(golden, oracle_backend, name):
    """State after every step of mlp.py:57-70 behind the fused norm, bit-exact; in the 8-bit trace a new outlier column arrives at call 1,
    so gate_proj's take-over from cache.new_ind (linear.py:298-315) IS exercised."""
    import g8_replay
    g = golden(name)
    worst = g8_replay.replay_walk(g, "cpu", lambda cache, M, KB: backend_oracle._plain(cache.q_xcache, M).view(np.uint8))
    assert worst <= 4e-3
    if int(g["bit"]) == 8:
        assert g["c0_up_ind"].tolist() == [7, 100] and g["c1_gate_ind"].tolist() == [7, 100, 201] and g["c1_new_ind"].tolist() == [201]


@pytest.mark.parametrize("name", ["g8_mlp_block_w8.npz", "g8_mlp_block_w4.npz"])
def test_g8_block_trace_through_forward(golden, oracle_backend, name):
    """MixLlamaMLP.forward (the multiply in gate_proj's epilogue instead of a separate pass) against the reference's y and layer state."""
    import g8_replay
    worst = g8_replay.replay_forward(golden(name), "cpu")
    assert worst <= 1e-2


def test_falcon_and_gptj_mlp_wrappers_match_their_composition(oracle_backend):
    """mixquant/modules/fused/mlp.py:8-32, :75-93: the wrappers ARE the sequence of operator calls the reference writes out - same calls, same
    flags (dense_4h_to_h with the unfused pre-pass), same activation modules - so their outputs equal the layers called by hand, bit for bit."""
    torch.manual_seed(5)
    K, I, M = 128, 256, 8
    a, b = torch.nn.Linear(K, I).half(), torch.nn.Linear(I, K).half()
    cache = MixLibCache(16, device="cpu")
    mk = lambda l, **kw: MixLinear_GEMM.from_linear(l, 8, cache=cache, dev="cpu", **kw)
    x = torch.randn(M, K).half()
    # (as in the reference's models, the first Linear's quantised input is left in the cache by the fused norm in front of it: fused/norm.py:21-33)
    def normed(first):
        norm = FasterTransformerRMSNorm(torch.ones(K), cache=cache)
        norm.next_layer = first
        return norm(x.clone())
    h_to_4h, h4_to_h = mk(a), mk(b)
    y = MixFalconMLP(h_to_4h, h4_to_h, cache)(normed(h_to_4h))
    h_to_4h2, h4_to_h2 = mk(a), mk(b)
    ref = h4_to_h2(torch.nn.GELU()(h_to_4h2(normed(h_to_4h2), cache)), cache, True)
    assert torch.equal(y, ref) and tuple(y.shape) == (M, K)
    xn = norm_ref(x)
    dense = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xn, a.weight.float(), a.bias.float())), b.weight.float(), b.bias.float())
    assert (y.float() - dense).abs().max() < 0.05 * dense.abs().max() + 0.05

    class Cfg:
        activation_function, resid_pdrop = "gelu_new", 0.0

    class Mod(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc_in, self.fc_out = mk(a), mk(b, weight_only=True, name="fc_out")     # the GPT-J policy: fc_out weight-only (utils/module.py:4-12)

    m = Mod()
    g = MixGPTJMLP(m, Cfg(), cache).eval()
    y2 = g(normed(m.fc_in))
    from transformers.activations import ACT2FN
    m2 = Mod()
    ref2 = m2.fc_out(ACT2FN["gelu_new"](m2.fc_in(normed(m2.fc_in), cache)), cache)
    assert torch.equal(y2, ref2) and tuple(y2.shape) == (M, K)
    from mixq_amd import MLPCache
    assert isinstance(MixGPTJMLP(m, Cfg()).MLPCache, MLPCache)            # (no model cache given: the reference's bare MLPCache)
