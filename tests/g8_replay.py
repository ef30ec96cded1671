"""Replay of the G8 fixtures (tests/golden/g8_mlp_block_w{8,4}.npz, recorded from the reference's own norm.py / mlp.py / linear.py by
oracle/gen_golden_g8.py) through mixq_amd's modules - shared by the host test (oracle backend, CPU) and the GPU tests (HIP backend).
Test infrastructure.

Per call the fixture holds the state the reference left after each step of mixquant/modules/fused/mlp.py:57-70 behind the fused norm
(norm.py:15-39): what the norm handed over, up_proj_'s result and its outlier bookkeeping, gate_proj_'s take-over of the new columns
(linear.py:298-315), the product, down_proj_'s result.  State (ind / weight_cache / cnt / add_outliers / forward_without_precondition_len /
new_ind / x_scale / q_xcache / activation_outliers / the in-place zeroed hidden) is compared bit-exactly, y within the tolerance given."""
import numpy as np
import torch

from mixq_amd import FasterTransformerRMSNorm, MixLibCache, MixLinear_GEMM, MixLlamaMLP


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


def build_block(g, dev, config=None):
    bit = int(g["bit"])
    I, K = g["up_weight"].shape
    up, gate, down = torch.nn.Linear(K, I, bias=False).half(), torch.nn.Linear(K, I, bias=False).half(), torch.nn.Linear(I, K, bias=False).half()
    up.weight.data.copy_(torch.from_numpy(g["up_weight"]))
    gate.weight.data.copy_(torch.from_numpy(g["gate_weight"]))
    down.weight.data.copy_(torch.from_numpy(g["down_weight"]))
    ls = torch.from_numpy(g["layer_scales"]) if "layer_scales" in g.files else None
    cache = MixLibCache(64, bit=bit, device=dev, config=config)
    mk = lambda l, b, s: MixLinear_GEMM.from_linear(l, b, cache=cache, layer_scales=s, dev=dev)
    up_q, gate_q, down_q = mk(up, bit, ls), mk(gate, bit, ls), mk(down, 8, None)
    norm = FasterTransformerRMSNorm(torch.from_numpy(g["norm_weight"]).to(dev), eps=float(g["eps"]), cache=cache)
    block = MixLlamaMLP(gate_q, down_q, up_q, cache)
    norm.next_layer = block.up_proj_                       # models/llama.py:20-22
    return norm, block, cache


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def check_layer(g, pre, layer):
    assert np.array_equal(_np(layer.ind).astype(np.int32), g[pre + "ind"]), pre + "ind"
    assert layer.cnt == int(g[pre + "cnt"]), pre + "cnt"
    assert layer.add_outliers == bool(g[pre + "add_outliers"]), pre + "add_outliers"
    assert layer.forward_without_precondition_len == int(g[pre + "fwpl"]), pre + "forward_without_precondition_len"
    if pre + "weight_cache" in g.files and layer.ind.numel():
        assert np.array_equal(bits(_np(layer.weight_cache)), bits(g[pre + "weight_cache"])), pre + "weight_cache"


def replay_walk(g, dev, unpack_q, config=None, tol=1e-2):
    """The block's forward step by step, as the generator walked the reference's: state after every step against the fixture.
    unpack_q(cache, M, KB) -> the plain uint8 view of cache.q_xcache (the GPU backend keeps it tile-major)."""
    norm, block, cache = build_block(g, dev, config)
    up_q, gate_q, down_q = block.up_proj_, block.gate_proj_, block.down_proj_
    for nm in ("up", "gate", "down"):
        lay = getattr(block, nm + "_proj_")
        assert np.array_equal(_np(lay.q_weight), g[nm + "_q_weight"]) and np.array_equal(bits(_np(lay.scale_col)), bits(g[nm + "_scale_col"]))
    K, I = up_q.in_features, up_q.out_features
    KB = K if up_q.bit == 8 else K // 2
    worst = 0.0
    for i in range(int(g["ncalls"])):
        x0 = torch.from_numpy(g[f"c{i}_x_in"].copy()).to(dev)
        x = x0.clone()
        M = x.numel() // K
        hidden = norm(x)
        assert torch.equal(x, x0), "the norm must not write its input"
        assert np.array_equal(bits(_np(cache.x_scale)[:M]), bits(g[f"c{i}_n_x_scale"])), f"call {i}: x_scale after the norm"
        assert np.array_equal(unpack_q(cache, M, KB), g[f"c{i}_n_q_xcache"].view(np.uint8)), f"call {i}: q_xcache after the norm"
        assert np.array_equal(bits(_np(hidden)), bits(g[f"c{i}_n_hidden"])), f"call {i}: normalised activation (outlier columns zeroed)"
        if f"c{i}_n_activation_outliers" in g.files and g[f"c{i}_n_activation_outliers"].shape[1]:
            assert np.array_equal(bits(_np(cache.activation_outliers)), bits(g[f"c{i}_n_activation_outliers"])), f"call {i}: outliers after the norm"
        up_out = up_q(hidden, block.MLPCache)
        assert tuple(cache.shape) == tuple(int(v) for v in g[f"c{i}_cache_shape"])
        assert np.array_equal(bits(_np(cache.x_scale)[:M]), bits(g[f"c{i}_u_x_scale"])), f"call {i}: x_scale after up_proj"
        assert np.array_equal(unpack_q(cache, M, KB), g[f"c{i}_u_q_xcache"].view(np.uint8)), f"call {i}: q_xcache after up_proj"
        assert np.array_equal(bits(_np(hidden)), bits(g[f"c{i}_u_hidden_after"])), f"call {i}: new outlier columns zeroed in place"
        if f"c{i}_u_activation_outliers" in g.files:
            assert np.array_equal(bits(_np(cache.activation_outliers)), bits(g[f"c{i}_u_activation_outliers"])), f"call {i}: outliers after up_proj"
        if f"c{i}_new_ind" in g.files:
            assert np.array_equal(_np(cache.new_ind).astype(np.int32), g[f"c{i}_new_ind"]), f"call {i}: cache.new_ind"
        d = np.abs(_np(up_out).astype(np.float32) - g[f"c{i}_u_y"].astype(np.float32)).max()
        assert d <= 4e-3, f"call {i}: up_proj |dy| = {d}"
        gate_out = gate_q.forward_without_preconditionFusedSilu(hidden, block.MLPCache)
        d = np.abs(_np(gate_out).astype(np.float32) - g[f"c{i}_g_y"].astype(np.float32)).max()
        assert d <= 4e-3, f"call {i}: gate_proj |dy| = {d}"
        gate_out *= up_out
        y = down_q(gate_out, None, True)
        d = float(np.abs(_np(y).astype(np.float32) - g[f"c{i}_y"].astype(np.float32)).max())
        worst = max(worst, d)
        assert d <= tol, f"call {i}: block |dy| = {d}"
        check_layer(g, f"c{i}_up_", up_q)
        check_layer(g, f"c{i}_gate_", gate_q)
        check_layer(g, f"c{i}_down_", down_q)
    return worst


def replay_forward(g, dev, config=None, tol=1e-2):
    """block(norm(x)) - the product route (gate / up as one launch once frozen, or two launches with the multiply in gate_proj's epilogue,
    down_proj's row maxima out of the same epilogue) - against the reference's y and final layer state."""
    norm, block, cache = build_block(g, dev, config)
    worst = 0.0
    for i in range(int(g["ncalls"])):
        x = torch.from_numpy(g[f"c{i}_x_in"].copy()).to(dev)
        y = block(norm(x))
        assert tuple(y.shape) == g[f"c{i}_y"].shape
        d = float(np.abs(_np(y).astype(np.float32) - g[f"c{i}_y"].astype(np.float32)).max())
        worst = max(worst, d)
        assert d <= tol, f"call {i}: block |dy| = {d}"
        check_layer(g, f"c{i}_up_", block.up_proj_)
        check_layer(g, f"c{i}_gate_", block.gate_proj_)
        check_layer(g, f"c{i}_down_", block.down_proj_)
    return worst
