#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON on seeded inputs.  TEST INFRASTRUCTURE ONLY.

Runs only in the build container (needs /root/reference, which does not exist on the GPU box); the fixtures it
writes are committed.  Nothing from the reference is copied: its modules `mixquant/modules/linear.py` and
`mixquant/Cache.py` are imported by file path, unmodified, with
  * `mixlib` / `EETQ` (absent native extensions) replaced by stand-ins: EETQ is never called on
    this path; the `mixlib` stand-in forwards to the CPU oracle, so G5 pins the reference's CONTROL FLOW (which
    native is called when, with what, and how `ind`/`cnt`/`add_outliers`/`weight_cache` evolve) while the arithmetic
    inside the natives stays "unpinned" as documented in oracle/mixq_oracle.c;
  * three monkey-patches that let the CUDA-only lines run on CPU: torch.cuda.get_device_capability -> (8, 0)
    (linear.py:86), Tensor.cuda -> clone (linear.py:116,128,133,142-143), Tensor.to('cuda') -> cpu (Cache.py:8-12).

Fixtures
  G1 pack_to_i4     : all 256 (lo,hi) nibble pairs + a random [16,64] matrix                       (bit-exact)
  G2 from_linear 8  : seeded nn.Linear(256,96,bias) fp16 -> q_weight, scale_col, bias               (bit-exact)
  G3 from_linear 4  : seeded nn.Linear(512,64) fp16 + layer_scales -> q_weight, weight_cache, ind, scale_col
  G4 FindOutliers   : seeded X with planted outlier columns -> ind
  G5 forward traces : per call ind / cnt / add_outliers / x_scale / q_xcache / y / mutated x, for
                      bit 8 unfused=True (+bias, new outlier on call 2), bit 8 unfused=False (caller-filled cache),
                      bit 8 without outliers, bit 4 (+ SiLU variant sharing the cache)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path[0] = ROOT
from oracle import oracle as O  # noqa: E402


def t2h(t):
    return t.detach().cpu().contiguous().numpy().copy()      # copy: CPU tensors share memory with reused cache buffers


def make_mixlib_standin():
    m = types.ModuleType("mixlib")
    m.calls = []

    def FindRowScale(x, x_scale, M, K, bit=8):
        m.calls.append("FindRowScale")
        q, s = O.find_row_scale(t2h(x.reshape(-1, K)[:M]), bit)
        x_scale[0:M] = torch.from_numpy(s).reshape(M, 1)
        return torch.from_numpy(q)

    def ExtractOutliersAndSetToZeros(ind, x):
        m.calls.append("ExtractOutliersAndSetToZeros")
        xn = t2h(x).copy()
        out = O.extract_outliers_zero(xn, t2h(ind))
        x.copy_(torch.from_numpy(xn))            # in-place mutation of the caller's tensor
        return torch.from_numpy(out)

    def _addend(addend, M, N):
        if tuple(addend.shape) == (M, N):
            return t2h(addend)
        return t2h(addend.reshape(-1)[: M * N].reshape(M, N))   # the native reads it as a dense [M,N]

    def _fused(act, bit):
        def f(q_x, q_w, x_scale, scale_col, addend, M, N, K):
            m.calls.append(("int8" if bit == 8 else "int4") + "FusedDequantize" + ("Silu" if act else ""))
            y = O.linear_fused(t2h(q_x), t2h(q_w), t2h(x_scale[0:M]), t2h(scale_col), addend=_addend(addend, M, N), act=act,
                               bit=bit)
            return torch.from_numpy(y)
        return f

    def unpack_int4_to_fp16(w, ind):
        m.calls.append("unpack_int4_to_fp16")
        return torch.from_numpy(O.unpack_i4_cols(t2h(w), t2h(ind)))

    m.FindRowScale = FindRowScale
    m.ExtractOutliersAndSetToZeros = ExtractOutliersAndSetToZeros
    m.int8FusedDequantize = _fused(0, 8)
    m.int8FusedDequantizeSilu = _fused(1, 8)
    m.int4FusedDequantize = _fused(0, 4)
    m.int4FusedDequantizeSilu = _fused(1, 4)
    m.unpack_int4_to_fp16 = unpack_int4_to_fp16
    return m


def load_reference():
    if not os.path.isdir(REF):
        raise SystemExit("/root/reference not present: fixtures can only be regenerated in the build container")
    sys.modules["mixlib"] = make_mixlib_standin()
    eetq = types.ModuleType("EETQ")
    eetq.quant_weights = eetq.preprocess_weights = eetq.w8_a16_gemm = None
    sys.modules["EETQ"] = eetq
    torch.cuda.get_device_capability = lambda *a, **k: (8, 0)
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()
    _to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(v, str) and v.startswith("cuda")) else v for v in a)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = "cpu"
        return _to(self, *a, **k)

    torch.Tensor.to = to

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    lin = load("ref_linear", os.path.join(REF, "mixquant/modules/linear.py"))
    cache = load("ref_cache", os.path.join(REF, "mixquant/Cache.py"))
    return lin, cache


def planted(M, K, cols, seed, scale=20.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).half()
    x[:, cols] *= scale
    return x


def trace_calls(layer, cache, xs, unfused, ref_mixlib, prefill=None, silu_layer=None):
    """Run layer.forward on each x (a fresh clone: the reference mutates it) and record the state after each call."""
    rec = {}
    for i, x0 in enumerate(xs):
        x = x0.clone()
        if prefill is not None:
            prefill(layer, cache, x)
        ref_mixlib.calls.clear()
        y = layer.forward(x, cache, unfused) if cache is not None else layer.forward(x, None, unfused)
        rec[f"c{i}_x_in"] = t2h(x0)
        rec[f"c{i}_x_after"] = t2h(x)
        rec[f"c{i}_y"] = t2h(y)
        rec[f"c{i}_ind"] = t2h(layer.ind).astype(np.int32)
        rec[f"c{i}_cnt"] = np.int32(layer.cnt)
        rec[f"c{i}_add_outliers"] = np.bool_(layer.add_outliers)
        M = x.reshape(-1, x.shape[-1]).shape[0]
        c = cache if cache is not None else layer.cache
        rec[f"c{i}_x_scale"] = t2h(c.x_scale[0:M])
        rec[f"c{i}_q_xcache"] = t2h(c.q_xcache)
        rec[f"c{i}_calls"] = np.array(ref_mixlib.calls)
        if layer.weight_cache is not None:
            rec[f"c{i}_weight_cache"] = t2h(layer.weight_cache)
        if c.activation_outliers is not None and layer.ind.shape[0]:
            rec[f"c{i}_activation_outliers"] = t2h(c.activation_outliers)
        if silu_layer is not None:
            ys = silu_layer.forward_without_preconditionFusedSilu(x0.clone(), c)
            rec[f"c{i}_y_silu"] = t2h(ys)
            rec[f"c{i}_silu_ind"] = t2h(silu_layer.ind).astype(np.int32)
    rec["ncalls"] = np.int32(len(xs))
    return rec


def main():
    os.makedirs(OUT, exist_ok=True)
    lin, cachemod = load_reference()
    ref_mixlib = sys.modules["mixlib"]

    # ---- G1 ----------------------------------------------------------------------------------------------
    pairs = torch.tensor([[lo, hi] for lo in range(-8, 8) for hi in range(-8, 8)], dtype=torch.int8)   # [256,2]
    g = torch.Generator().manual_seed(1)
    rnd = torch.randint(-8, 8, (16, 64), generator=g, dtype=torch.int8)
    np.savez(os.path.join(OUT, "g1_pack_i4.npz"), pairs=t2h(pairs), pairs_packed=t2h(lin.pack_to_i4(pairs)), rnd=t2h(rnd),
             rnd_packed=t2h(lin.pack_to_i4(rnd)), two_compl_in=np.arange(-8, 8, dtype=np.int8),
             two_compl_out=t2h(lin.two_compl(torch.arange(-8, 8, dtype=torch.int8), 4)))

    # ---- G2 ----------------------------------------------------------------------------------------------
    torch.manual_seed(0)
    l8 = torch.nn.Linear(256, 96, bias=True).half()
    W8 = l8.weight.data.clone()
    cache8 = cachemod.MixLibCache(64)
    q8 = lin.MixLinear_GEMM.from_linear(l8, bit=8, weight_only=False, init_only=False, cache=cache8, dev="cpu", name="g2")
    np.savez(os.path.join(OUT, "g2_from_linear_w8.npz"), weight=t2h(W8), bias_in=t2h(l8.bias.data), q_weight=t2h(q8.q_weight),
             scale_col=t2h(q8.scale_col), bias=t2h(q8.bias))

    # ---- G3 ----------------------------------------------------------------------------------------------
    torch.manual_seed(3)
    l4 = torch.nn.Linear(512, 64, bias=False).half()
    W4 = l4.weight.data.clone()
    g = torch.Generator().manual_seed(4)
    layer_scales = torch.rand(512, generator=g) * 5 + 0.1
    cache4 = cachemod.MixLibCache(64, bit=4)
    q4 = lin.MixLinear_GEMM.from_linear(l4, bit=4, weight_only=False, init_only=False, cache=cache4, layer_scales=layer_scales,
                                        dev="cpu", name="g3")
    np.savez(os.path.join(OUT, "g3_from_linear_w4.npz"), weight=t2h(W4), layer_scales=t2h(layer_scales),
             q_weight=t2h(q4.q_weight), scale_col=t2h(q4.scale_col), weight_cache=t2h(q4.weight_cache),
             ind=t2h(q4.ind).astype(np.int32))

    # ---- G4 ----------------------------------------------------------------------------------------------
    xg4 = planted(24, 256, [7, 100, 201], seed=5)
    xg4[3, 55] = 6.0      # exactly sigma: NOT an outlier (strict >)
    xg4[4, 56] = 6.004    # rounds to the next fp16 above 6 -> outlier
    np.savez(os.path.join(OUT, "g4_find_outliers.npz"), x=t2h(xg4), sigma=np.float32(6.0), ind=t2h(q8.FindOutliers(xg4)))

    # ---- G5a: bit 8, unfused=True, bias, outliers [7,100,201] on call 1 and a NEW column 33 on call 2 --------
    torch.manual_seed(0)
    l8 = torch.nn.Linear(256, 96, bias=True).half()
    cache = cachemod.MixLibCache(64)
    layer = lin.MixLinear_GEMM.from_linear(l8, bit=8, cache=cache, dev="cpu", name="g5a")
    xs = [planted(32, 256, [7, 100, 201], seed=10), planted(32, 256, [7, 100, 201, 33], seed=11),
          planted(32, 256, [7, 100, 201, 33], seed=12), planted(32, 256, [7, 100, 201, 33, 150], seed=13)]
    np.savez(os.path.join(OUT, "g5a_forward_w8_unfused.npz"), **trace_calls(layer, None, xs, True, ref_mixlib))

    # ---- G5b: bit 8, unfused=False: the caller (the fused norm, norm.py:24-33) fills the cache first ----------
    torch.manual_seed(0)
    l8 = torch.nn.Linear(256, 96, bias=False).half()
    cache = cachemod.MixLibCache(64)
    layer = lin.MixLinear_GEMM.from_linear(l8, bit=8, cache=cache, dev="cpu", name="g5b")

    def prefill(layer, cache, x):
        inputs = x.reshape(-1, x.shape[-1])
        if layer.ind.shape[0]:
            cache.activation_outliers = ref_mixlib.ExtractOutliersAndSetToZeros(layer.ind, inputs)
        cache.q_xcache = ref_mixlib.FindRowScale(inputs, cache.x_scale, inputs.shape[0], layer.in_features, layer.bit)

    xs3 = [x.reshape(2, 16, 256) for x in xs[:3]]
    np.savez(os.path.join(OUT, "g5b_forward_w8_fused_cache.npz"), **trace_calls(layer, cache, xs3, False, ref_mixlib, prefill))

    # ---- G5c: bit 8, no outliers at all (zeros addend path) --------------------------------------------------
    torch.manual_seed(0)
    l8 = torch.nn.Linear(256, 96, bias=True).half()
    cache = cachemod.MixLibCache(64)
    layer = lin.MixLinear_GEMM.from_linear(l8, bit=8, cache=cache, dev="cpu", name="g5c")
    g = torch.Generator().manual_seed(20)
    xs_plain = [torch.randn(32, 256, generator=g).half() for _ in range(3)]
    np.savez(os.path.join(OUT, "g5c_forward_w8_no_outliers.npz"), **trace_calls(layer, None, xs_plain, True, ref_mixlib))

    # ---- G5d: bit 4 (128 static fp columns) + the SiLU twin sharing the cache (mlp.py:61-62) -----------------
    torch.manual_seed(3)
    up = torch.nn.Linear(512, 64, bias=False).half()
    gate = torch.nn.Linear(512, 64, bias=False).half()
    cache = cachemod.MixLibCache(64, bit=4)
    up_q = lin.MixLinear_GEMM.from_linear(up, bit=4, cache=cache, layer_scales=layer_scales, dev="cpu", name="up")
    gate_q = lin.MixLinear_GEMM.from_linear(gate, bit=4, cache=cache, layer_scales=layer_scales, dev="cpu", name="gate")
    g = torch.Generator().manual_seed(21)
    xs4 = [torch.randn(16, 512, generator=g).half() for _ in range(2)]
    rec = trace_calls(up_q, cache, xs4, True, ref_mixlib, silu_layer=gate_q)
    rec["up_weight"], rec["gate_weight"], rec["layer_scales"] = t2h(up.weight.data), t2h(gate.weight.data), t2h(layer_scales)
    np.savez(os.path.join(OUT, "g5d_forward_w4_silu.npz"), **rec)

    # ---- G6: checkpoint layout: state_dict keys / shapes / dtypes of the reference module per flavour, and the
    #      layer policy tables of utils/module.py ------------------------------------------
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("ref_utils_module", os.path.join(REF, "mixquant", "utils", "module.py"))
    um = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(um)

    def layout(m):
        return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}

    lw = torch.nn.Linear(256, 96, bias=True).half()
    wo = lin.MixLinear_GEMM.from_linear(lw, bit=8, weight_only=True, init_only=True, cache=cache8, dev="cpu", name="g6wo")
    g6 = {
        "w8_bias": layout(q8),
        "w4_nobias": layout(q4),
        "weight_only_w8_bias": layout(wo),
        "eightbit_only_name": list(um.eightbit_only_name),
        "weight_only_map": {k: list(v) for k, v in um.weight_only_map.items()},
    }
    with open(os.path.join(OUT, "g6_checkpoint_layout.json"), "w") as f:
        json.dump(g6, f, indent=1, sort_keys=True)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
