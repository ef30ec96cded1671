"""Pins the CPU oracle (and the operator's torch-level host code) against vectors produced by RUNNING THE REFERENCE
(oracle/gen_golden.py; fixtures under tests/golden/).  All comparisons here are bit-exact."""
import numpy as np
import pytest
import torch

from mixq_amd import MixLibCache, MixLinear_GEMM, pack_to_i4, two_compl


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


def test_g1_pack_to_i4(golden, oracle):
    g = golden("g1_pack_i4.npz")
    assert np.array_equal(oracle.pack_i4(g["pairs"]), g["pairs_packed"])
    assert np.array_equal(oracle.pack_i4(g["rnd"]), g["rnd_packed"])
    # the operator module's own helpers (API mirror of linear.py:12-18)
    assert np.array_equal(pack_to_i4(torch.from_numpy(g["pairs"])).numpy(), g["pairs_packed"])
    assert np.array_equal(pack_to_i4(torch.from_numpy(g["rnd"])).numpy(), g["rnd_packed"])
    assert np.array_equal(two_compl(torch.from_numpy(g["two_compl_in"]), 4).numpy(), g["two_compl_out"])
    # and the inverse
    assert np.array_equal(oracle.unpack_i4_all(g["rnd_packed"]), g["rnd"])


def test_g2_from_linear_w8(golden, oracle):
    g = golden("g2_from_linear_w8.npz")
    q, s = oracle.quant_weight_w8(g["weight"])
    assert np.array_equal(q, g["q_weight"])
    assert np.array_equal(bits(s), bits(g["scale_col"]))
    lin = torch.nn.Linear(256, 96, bias=True).half()
    lin.weight.data.copy_(torch.from_numpy(g["weight"]))
    lin.bias.data.copy_(torch.from_numpy(g["bias_in"]))
    w_before = lin.weight.data.clone()
    ql = MixLinear_GEMM.from_linear(lin, 8, cache=MixLibCache(64, device="cpu"), dev="cpu")
    assert np.array_equal(ql.q_weight.numpy(), g["q_weight"])
    assert np.array_equal(bits(ql.scale_col.numpy()), bits(g["scale_col"]))
    assert np.array_equal(bits(ql.bias.numpy()), bits(g["bias"]))
    assert torch.equal(lin.weight.data, w_before), "from_linear must not destroy the caller's weight"
    assert ql.q_weight.dtype == torch.int8 and tuple(ql.q_weight.shape) == (96, 256)
    assert ql.scale_col.dtype == torch.float16 and tuple(ql.scale_col.shape) == (1, 96)
    assert ql.ind.dtype == torch.int32 and ql.ind.numel() == 0 and ql.weight_cache is None


def test_g3_from_linear_w4(golden, oracle):
    g = golden("g3_from_linear_w4.npz")
    ind = torch.sort(torch.from_numpy(g["layer_scales"]))[1][-128:].numpy().astype(np.int32)
    assert np.array_equal(ind, g["ind"])
    q, s, wc = oracle.quant_weight_w4(g["weight"], ind)
    assert np.array_equal(q, g["q_weight"])
    assert np.array_equal(bits(s), bits(g["scale_col"]))
    assert np.array_equal(bits(wc), bits(g["weight_cache"]))
    lin = torch.nn.Linear(512, 64, bias=False).half()
    lin.weight.data.copy_(torch.from_numpy(g["weight"]))
    ql = MixLinear_GEMM.from_linear(lin, 4, cache=MixLibCache(64, bit=4, device="cpu"), layer_scales=torch.from_numpy(g["layer_scales"]),
                                    dev="cpu")
    assert np.array_equal(ql.q_weight.numpy(), g["q_weight"])
    assert np.array_equal(bits(ql.scale_col.numpy()), bits(g["scale_col"]))
    assert np.array_equal(bits(ql.weight_cache.numpy()), bits(g["weight_cache"]))
    assert np.array_equal(ql.ind.numpy(), g["ind"])
    assert ql.q_weight.dtype == torch.uint8 and tuple(ql.q_weight.shape) == (64, 256)
    # buffers the reference registers for 4-bit layers (checkpoint keys)
    assert {"q_weight", "scale_col", "weight_cache", "ind"} <= set(dict(ql.named_buffers()).keys())
    # unpack(pack) restores the quantised integers of the non-fp columns
    cols = np.arange(512, dtype=np.int32)
    full = oracle.unpack_i4_cols(g["q_weight"], cols).astype(np.float32)
    assert np.array_equal(full.astype(np.int8), oracle.unpack_i4_all(g["q_weight"]))
    assert np.all(full[:, ind] == 0)


def test_g4_find_outliers(golden, oracle):
    g = golden("g4_find_outliers.npz")
    got = oracle.find_outliers(g["x"], float(g["sigma"]))
    assert np.array_equal(got, g["ind"])
    assert 55 not in got and 56 in got          # exactly sigma is not an outlier (strict >), the next fp16 is


@pytest.mark.parametrize("name,bit", [("g5a_forward_w8_unfused.npz", 8), ("g5c_forward_w8_no_outliers.npz", 8),
                                       ("g5d_forward_w4_silu.npz", 4)])
def test_g5_quantise_step_replays(golden, oracle, name, bit):
    """Per call: extracting the recorded `ind` from the recorded input and quantising reproduces the recorded
    x_scale / q_xcache / mutated x / outlier matrix (the natives' contracts as restated in the oracle)."""
    g = golden(name)
    for i in range(int(g["ncalls"])):
        x = g[f"c{i}_x_in"].reshape(-1, g[f"c{i}_x_in"].shape[-1]).copy()
        ind = g[f"c{i}_ind"]
        xo = oracle.extract_outliers_zero(x, ind)
        q, s = oracle.find_row_scale(x, bit)
        assert np.array_equal(bits(x), bits(g[f"c{i}_x_after"].reshape(x.shape)))
        assert np.array_equal(q, g[f"c{i}_q_xcache"])
        assert np.array_equal(bits(s), bits(g[f"c{i}_x_scale"].reshape(-1)))
        if ind.size and f"c{i}_activation_outliers" in g.files:
            assert np.array_equal(bits(xo), bits(g[f"c{i}_activation_outliers"]))


def test_oracle_gemm_is_exact_integer(oracle):
    rng = np.random.default_rng(0)
    qx = rng.integers(-127, 128, (7, 192), dtype=np.int8)
    qw = rng.integers(-127, 128, (12, 192), dtype=np.int8)
    assert np.array_equal(oracle.gemm_i8(qx, qw), qx.astype(np.int64) @ qw.astype(np.int64).T)


def test_oracle_fused_vs_dequant_reference(oracle):
    """The integer-path restatement agrees with the independent fp64 Linear over dequantised operands (the north_star
    gate) to fp16 rounding."""
    rng = np.random.default_rng(1)
    M, N, K = 16, 40, 256
    x = (rng.standard_normal((M, K)) * 1.0).astype(np.float16)
    ind = np.array([3, 77, 200], dtype=np.int32)
    x[:, ind] *= 20
    w = (rng.standard_normal((N, K)) / 16).astype(np.float16)
    qw, sw = oracle.quant_weight_w8(w)
    xz = x.copy()
    xo = oracle.extract_outliers_zero(xz, ind)
    qx, sx = oracle.find_row_scale(xz, 8)
    wo = oracle.dequant_weight_cols(qw, sw, ind, 8)
    bias = rng.standard_normal(N).astype(np.float16)
    y = oracle.linear_fused(qx, qw, sx, sw, xo=xo, wo=wo, bias=bias).astype(np.float64)
    ref = oracle.linear_dequant_ref(qx, qw, sx, sw, xo=xo, ind=ind, bias=bias, wo=wo)
    assert np.abs(y - ref).max() <= 1e-2


def test_row_scale_multiplication_equals_the_division_for_every_fp16_maximum():
    """The kernels compute x_scale = fp16(amax * fp32(1 / qmax)) (mixq_amd/csrc/common.h: mixq_row_scale); the convention (include/mixq_hip.h,
    oracle find_row_scale) is fp16(amax / qmax) with an fp32 IEEE division.  The two agree for every finite fp16 maximum and both widths."""
    import numpy as np
    a = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float32)
    for qmax in (127.0, 7.0):
        div = (a / np.float32(qmax)).astype(np.float16)
        mul = (a * (np.float32(1.0) / np.float32(qmax))).astype(np.float16)
        assert np.array_equal(div.view(np.uint16), mul.view(np.uint16))


def test_fullsize_fixture_is_what_the_oracle_computes(oracle):
    """tests/golden/g7 (the metric shape, committed so that the GPU box needs no oracle for it) against a fresh run of its generator's recipe."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("gen_fullsize_fixture", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "gen_fullsize_fixture.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_fullsize_512x4096x11008.npz"))
    w, x, ind = gen.inputs()
    assert np.array_equal(ind, f["ind"]) and list(f["rows"]) == gen.ROWS
    qw, sw = oracle.quant_weight_w8(w)
    assert np.array_equal(sw.reshape(-1).view(np.uint16), f["scale_col"].view(np.uint16))
    assert np.array_equal(qw.astype(np.int32).sum(axis=1), f["q_weight_rowsum"])
    xz = x.copy()
    xo = oracle.extract_outliers_zero(xz, ind)
    qx, sx = oracle.find_row_scale(xz, 8)
    assert np.array_equal(sx.view(np.uint16), f["x_scale"].view(np.uint16)) and np.array_equal(qx[gen.ROWS], f["q_x"])
    assert np.array_equal(xo[gen.ROWS].view(np.uint16), f["x_out"].view(np.uint16))
