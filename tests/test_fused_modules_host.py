"""Host logic of the modules either side of the path on the oracle backend: the fused
RMSNorm hands the next linear exactly what its own unfused pre-pass would have produced, and the MLP wrapper calls the
operator the way mixquant/modules/fused/mlp.py:57-70 does."""
import numpy as np
import pytest
import torch

import backend_oracle
import mixq_amd.fused as F
import mixq_amd.linear as L
from mixq_amd import FasterTransformerRMSNorm, MixLibCache, MixLinear_GEMM, MixLlamaMLP
from oracle import oracle as O


@pytest.fixture()
def oracle_backend():
    p1, p2 = L.set_backend(backend_oracle), F.set_backend(backend_oracle)
    backend_oracle.calls.clear()
    yield backend_oracle
    L.set_backend(p1)
    F.set_backend(p2)


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
