#!/usr/bin/env python3
"""G8: the norm -> up_proj -> gate_proj (SiLU, shared activation) -> product -> down_proj flow of a Llama MLP block, recorded by RUNNING
THE REFERENCE'S OWN PYTHON (`mixquant/modules/fused/norm.py`, `fused/mlp.py`, `modules/linear.py`, `Cache.py`, imported by file path,
unmodified) on seeded inputs.  TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference); writes
tests/golden/g8_mlp_block_w8.npz and g8_mlp_block_w4.npz.

What it pins that G1-G7 do not: the hand-over of the quantised activation from the fused norm to `up_proj_` (norm.py:24-33, wiring
models/llama.py:20-22), and `gate_proj_`'s take-over of NEW outlier columns from `cache.new_ind` (linear.py:298-315) - the 8-bit trace
plants a new column at call 1, while up_proj's search is still running, so that block IS entered (G5d is 4-bit with static columns and
never enters it).  The natives (`mixlib.layernorm_forward_cuda*`, `FindRowScale`, `ExtractOutliersAndSetToZeros`, `int8/int4FusedDequantize
[Silu]`) are the oracle stand-ins of gen_golden.py plus the three norm entry points below: control flow and state are the reference's,
in-native rounding is the oracle's.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden as G  # noqa: E402
from oracle import oracle as O  # noqa: E402

REF, OUT, t2h = G.REF, G.OUT, G.t2h


def add_norm_standins(m):
    """mixlib.layernorm_forward_cuda / _extract_outliers / _extract_outliers_int4 (call sites norm.py:21-33) on the oracle."""
    def layernorm_forward_cuda(x, weight, out, eps):
        m.calls.append("layernorm_forward_cuda")
        K = x.shape[-1]
        out.reshape(-1, K).copy_(torch.from_numpy(O.rmsnorm(t2h(x.reshape(-1, K)), t2h(weight), eps)))

    def _extract(bit, name):
        def f(x, weight, out, eps, ind, x_scale):
            m.calls.append(name)
            K = x.shape[-1]
            x2 = t2h(x.reshape(-1, K))
            M = x2.shape[0]
            y, xo, q, s = O.rmsnorm_quant(x2, t2h(weight), eps, t2h(ind).astype(np.int32), bit)
            out.reshape(-1, K).copy_(torch.from_numpy(y))
            x_scale[0:M] = torch.from_numpy(s).reshape(M, 1)
            return torch.from_numpy(xo), torch.from_numpy(q)
        return f

    m.layernorm_forward_cuda = layernorm_forward_cuda
    m.layernorm_forward_cuda_extract_outliers = _extract(8, "layernorm_forward_cuda_extract_outliers")
    m.layernorm_forward_cuda_extract_outliers_int4 = _extract(4, "layernorm_forward_cuda_extract_outliers_int4")


def load_block_modules(cachemod):
    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    pkg = types.ModuleType("mixquant")
    pkg.__path__ = []
    sys.modules["mixquant"] = pkg
    sys.modules["mixquant.Cache"] = cachemod               # mlp.py:4 imports MixLibCache / MLPCache from there
    norm = load("ref_norm", os.path.join(REF, "mixquant/modules/fused/norm.py"))
    mlp = load("ref_mlp", os.path.join(REF, "mixquant/modules/fused/mlp.py"))
    return norm, mlp


def layer_state(prefix, layer, rec):
    rec[prefix + "ind"] = t2h(layer.ind).astype(np.int32)
    rec[prefix + "cnt"] = np.int32(layer.cnt)
    rec[prefix + "add_outliers"] = np.bool_(layer.add_outliers)
    rec[prefix + "fwpl"] = np.int32(layer.forward_without_precondition_len)
    if layer.weight_cache is not None:
        rec[prefix + "weight_cache"] = t2h(layer.weight_cache)


def record_block(lin, cachemod, normmod, mlpmod, ref_mixlib, bit, K, I, M3, xs, layer_scales=None, seed=0):
    torch.manual_seed(seed)
    up, gate, down = (torch.nn.Linear(K, I, bias=False).half(), torch.nn.Linear(K, I, bias=False).half(),
                      torch.nn.Linear(I, K, bias=False).half())
    g = torch.Generator().manual_seed(seed + 100)
    norm_w = (1 + 0.1 * torch.randn(K, generator=g)).half()
    rec = {"up_weight": t2h(up.weight.data), "gate_weight": t2h(gate.weight.data), "down_weight": t2h(down.weight.data),
           "norm_weight": t2h(norm_w), "eps": np.float32(1e-6), "bit": np.int32(bit)}
    if layer_scales is not None:
        rec["layer_scales"] = t2h(layer_scales)
    cache = cachemod.MixLibCache(64, bit=bit)
    mk = lambda l, b, name, ls: lin.MixLinear_GEMM.from_linear(l, bit=b, cache=cache, layer_scales=ls, dev="cpu", name=name)
    up_q, gate_q = mk(up, bit, "up", layer_scales), mk(gate, bit, "gate", layer_scales)
    down_q = mk(down, 8, "down", None)                      # (o_proj / down_proj stay 8-bit: utils/module.py:2)
    norm = normmod.FasterTransformerRMSNorm(norm_w, eps=1e-6, cache=cache)
    block = mlpmod.MixLlamaMLP(gate_q, down_q, up_q, cache)
    norm.next_layer = block.up_proj_                        # models/llama.py:20-22
    for i, x0 in enumerate(xs):
        ref_mixlib.calls.clear()
        x = x0.clone()
        hidden = norm(x)
        M = x.reshape(-1, K).shape[0]
        rec[f"c{i}_x_in"] = t2h(x0)
        assert torch.equal(x, x0)                           # (the norm does not write its input)
        rec[f"c{i}_n_x_scale"] = t2h(cache.x_scale[0:M])    # the state the norm hands to up_proj_
        rec[f"c{i}_n_q_xcache"] = t2h(cache.q_xcache)
        rec[f"c{i}_n_hidden"] = t2h(hidden)
        if cache.activation_outliers is not None:
            rec[f"c{i}_n_activation_outliers"] = t2h(cache.activation_outliers)
        # the block's forward, step by step as mlp.py:57-70 runs it, so the intermediate state can be recorded
        up_out = block.up_proj_(hidden, block.MLPCache)
        rec[f"c{i}_u_y"] = t2h(up_out)
        rec[f"c{i}_u_x_scale"] = t2h(cache.x_scale[0:M])
        rec[f"c{i}_u_q_xcache"] = t2h(cache.q_xcache)
        rec[f"c{i}_u_hidden_after"] = t2h(hidden)           # new outlier columns are zeroed in place (linear.py:205)
        if cache.activation_outliers is not None and up_q.ind.shape[0]:
            rec[f"c{i}_u_activation_outliers"] = t2h(cache.activation_outliers)
        if getattr(cache, "new_ind", None) is not None:
            rec[f"c{i}_new_ind"] = t2h(cache.new_ind).astype(np.int32)
        rec[f"c{i}_cache_shape"] = np.array(tuple(cache.shape), dtype=np.int32)
        gate_out = block.gate_proj_.forward_without_preconditionFusedSilu(hidden, block.MLPCache)
        rec[f"c{i}_g_y"] = t2h(gate_out)
        gate_out *= up_out
        rec[f"c{i}_prod"] = t2h(gate_out)
        y = block.down_proj_(gate_out, None, True)
        rec[f"c{i}_y"] = t2h(y)
        rec[f"c{i}_d_x_scale"] = t2h(cache.x_scale[0:M])
        rec[f"c{i}_d_q_xcache"] = t2h(cache.q_xcache)
        rec[f"c{i}_calls"] = np.array([str(c) for c in ref_mixlib.calls])
        layer_state(f"c{i}_up_", up_q, rec)
        layer_state(f"c{i}_gate_", gate_q, rec)
        layer_state(f"c{i}_down_", down_q, rec)
        # ... and the same input through block.forward of a twin built from the same weights must give the same y (the step-by-step
        # walk above IS mlp.py:57-70; this guards the transcription)
    rec["ncalls"] = np.int32(len(xs))
    rec["up_q_weight"], rec["gate_q_weight"], rec["down_q_weight"] = t2h(up_q.q_weight), t2h(gate_q.q_weight), t2h(down_q.q_weight)
    rec["up_scale_col"], rec["gate_scale_col"], rec["down_scale_col"] = t2h(up_q.scale_col), t2h(gate_q.scale_col), t2h(down_q.scale_col)
    return rec, (norm, block, cache)


def check_forward_equals_walk(lin, cachemod, normmod, mlpmod, rec, bit, K, I, xs, layer_scales, seed):
    """The unmodified MixLlamaMLP.forward on a twin block: same y per call as the recorded step-by-step walk."""
    torch.manual_seed(seed)
    up, gate, down = (torch.nn.Linear(K, I, bias=False).half(), torch.nn.Linear(K, I, bias=False).half(),
                      torch.nn.Linear(I, K, bias=False).half())
    g = torch.Generator().manual_seed(seed + 100)
    norm_w = (1 + 0.1 * torch.randn(K, generator=g)).half()
    cache = cachemod.MixLibCache(64, bit=bit)
    mk = lambda l, b, ls: lin.MixLinear_GEMM.from_linear(l, bit=b, cache=cache, layer_scales=ls, dev="cpu")
    block = mlpmod.MixLlamaMLP(mk(gate, bit, layer_scales), mk(down, 8, None), mk(up, bit, layer_scales), cache)
    norm = normmod.FasterTransformerRMSNorm(norm_w, eps=1e-6, cache=cache)
    norm.next_layer = block.up_proj_
    for i, x0 in enumerate(xs):
        y = block(norm(x0.clone()))
        assert np.array_equal(t2h(y).view(np.uint16), rec[f"c{i}_y"].view(np.uint16)), f"call {i}: forward() and the recorded walk differ"


def main():
    os.makedirs(OUT, exist_ok=True)
    lin, cachemod = G.load_reference()
    ref_mixlib = sys.modules["mixlib"]
    add_norm_standins(ref_mixlib)
    normmod, mlpmod = load_block_modules(cachemod)

    # ---- 8-bit: columns {7, 100} hot from call 0, column 201 from call 1 (search still running: cnt 1 -> 2), frozen from call 2 on;
    #      3-D input [2, 8, K] so cache.shape is pinned as well ------------------------------------------------------------------
    K, I = 256, 192
    xs = []
    for i, cols in enumerate([[7, 100], [7, 100, 201], [7, 100, 201], [7, 100, 201]]):
        xs.append(G.planted(16, K, cols, seed=80 + i, scale=25.0).reshape(2, 8, K))
    rec, _ = record_block(lin, cachemod, normmod, mlpmod, ref_mixlib, 8, K, I, None, xs, seed=8)
    check_forward_equals_walk(lin, cachemod, normmod, mlpmod, rec, 8, K, I, xs, None, 8)
    assert rec["c0_up_ind"].tolist() == [7, 100] and rec["c1_up_ind"].tolist() == [7, 100, 201] and rec["c1_gate_ind"].tolist() == [7, 100, 201]
    assert int(rec["c0_gate_fwpl"]) == 2 and int(rec["c1_gate_fwpl"]) == 3 and rec["c1_new_ind"].tolist() == [201]
    assert bool(rec["c0_up_add_outliers"]) and not bool(rec["c1_up_add_outliers"])
    np.savez_compressed(os.path.join(OUT, "g8_mlp_block_w8.npz"), **rec)

    # ---- 4-bit gate / up (128 static fp16 columns from layer_scales, linear.py:123-143), 8-bit down ------------------------------
    K4, I4 = 512, 128
    g = torch.Generator().manual_seed(4)
    layer_scales = torch.rand(K4, generator=g) * 5 + 0.1
    g = torch.Generator().manual_seed(91)
    xs4 = [torch.randn(2, 8, K4, generator=g).half() for _ in range(3)]
    rec4, _ = record_block(lin, cachemod, normmod, mlpmod, ref_mixlib, 4, K4, I4, None, xs4, layer_scales=layer_scales, seed=9)
    check_forward_equals_walk(lin, cachemod, normmod, mlpmod, rec4, 4, K4, I4, xs4, layer_scales, 9)
    np.savez_compressed(os.path.join(OUT, "g8_mlp_block_w4.npz"), **rec4)
    for f in ("g8_mlp_block_w8.npz", "g8_mlp_block_w4.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))
    for i in range(4):
        print("w8 call", i, "up.ind", rec[f"c{i}_up_ind"].tolist(), "gate.ind", rec[f"c{i}_gate_ind"].tolist(), "gate.fwpl", int(rec[f"c{i}_gate_fwpl"]),
              "cnt", int(rec[f"c{i}_up_cnt"]), "down.ind", rec[f"c{i}_down_ind"].tolist(), "calls", list(rec[f"c{i}_calls"]))


if __name__ == "__main__":
    main()
