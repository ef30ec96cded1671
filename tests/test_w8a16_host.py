"""SURVEY.md §8f row 4, host side on CPU: EETQ-style weight quantisation, the weight-only operator flow and its
checkpoint layout, with the oracle standing in for the kernels (tests/backend_oracle.py).  Parity for this row is
UNPINNED (EETQ is absent from the reference tree): the oracle restates the published FasterTransformer rule."""
import numpy as np
import pytest
from conftest import swap_backend
import torch
import torch.nn as nn

import backend_oracle
from mixq_amd import MixLibCache, MixLinear_GEMM, eetq, linear as linmod
from mixq_amd import checkpoint as ck
from oracle import oracle as O


@pytest.fixture(autouse=True)
def _oracle_backend():
    prev = swap_backend(linmod, backend_oracle)
    swap_backend(eetq, backend_oracle)
    backend_oracle.calls.clear()
    yield
    swap_backend(linmod, prev)
    from mixq_amd import mixlib
    swap_backend(eetq, mixlib)


@pytest.mark.parametrize("K,N,seed", [(64, 16, 0), (256, 96, 1), (512, 40, 2)])
def test_quant_weights_matches_oracle(K, N, seed):
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(K, N, generator=g) * 0.05).half()
    w[:, 3] = 0                                   # all-zero output channel
    w[5, 7] = w[:, 7].abs().max() * 2             # positive maximum -> 128 clips to 127
    w[6, 8] = -w[:, 8].abs().max() * 2            # negative maximum -> exactly -128
    q, s = eetq.quant_weights(w, torch.int8, False)
    qo, so = O.quant_weight_w8a16(w.numpy())
    assert q.dtype == torch.int8 and q.shape == (K, N) and s.dtype == torch.float16 and s.shape == (N,)
    assert np.array_equal(q.numpy(), qo) and np.array_equal(s.numpy().view(np.uint16), so.view(np.uint16))
    assert q[:, 3].abs().max() == 0 and s[3] == 0
    assert q[5, 7] == 127 and q[6, 8] == -128
    # dequantised weights are within half a quantisation step (+ the fp16 rounding of the stored scale, <= 128 * 2^-11
    # steps; one whole step where the +128 clip bites)
    err = (q.float() * s.float() - w.float()).abs()
    bound = torch.where(q == 127, s.float() * 1.07, s.float() * 0.57) + 1e-7     # a column's positive maximum maps to 128
    assert (err <= bound).all()


def test_quant_weights_rejects_other_types():
    with pytest.raises(NotImplementedError):
        eetq.quant_weights(torch.zeros(4, 4).half(), torch.quint4x2, False)


@pytest.mark.parametrize("bias", [False, True])
def test_weight_only_operator_flow(bias):
    torch.manual_seed(0)
    K, N, M = 256, 96, 24
    lin = nn.Linear(K, N, bias=bias).half()
    cache = MixLibCache(64, device="cpu")
    q = MixLinear_GEMM.from_linear(lin, bit=8, weight_only=True, cache=cache, dev="cpu", name="fc_out")
    assert q.weight_only and q.q_weight.shape == (K, N) and q.q_weight.dtype == torch.int8 and q.scale_col.shape == (N,)
    assert set(q.state_dict()) == ({"q_weight", "scale_col", "bias"} if bias else {"q_weight", "scale_col"})
    qo, so = O.quant_weight_w8a16(lin.weight.data.t().contiguous().numpy())
    assert np.array_equal(q.q_weight.numpy(), qo) and np.array_equal(q.scale_col.numpy().view(np.uint16), so.view(np.uint16))
    x = torch.randn(2, M // 2, K).half()
    y = q(x)
    assert y.shape == (2, M // 2, N) and y.dtype == torch.float16 and cache.shape == (2, M // 2, N)
    assert backend_oracle.calls == ["PackW8A16", "W8A16Linear"]
    q(x)
    assert backend_oracle.calls == ["PackW8A16", "W8A16Linear", "W8A16Linear"]          # packed copy is cached
    ref = torch.nn.functional.linear(x.float(), (q.q_weight.float() * q.scale_col.float()).t(),
                                     None if not bias else q.bias.float())
    assert (y.float() - ref).abs().max() <= 1e-2
    # against the unquantised layer: int8 weight error only
    full = torch.nn.functional.linear(x.float(), lin.weight.float(), None if not bias else lin.bias.float())
    assert (y.float() - full).abs().max() < 5e-2


def test_eetq_surface_w8_a16_gemm():
    torch.manual_seed(1)
    w = (torch.randn(128, 32) * 0.1).half()
    q, s = eetq.quant_weights(w, torch.int8, False)
    assert torch.equal(eetq.unprocess_weights(eetq.preprocess_weights(q)), q)
    x = torch.randn(3, 5, 128).half()
    y = eetq.w8_a16_gemm(x, q, s)
    assert y.shape == (3, 5, 32)
    assert np.array_equal(y.reshape(-1, 32).numpy(), O.w8a16_linear(x.reshape(-1, 128).numpy(), q.numpy(), s.numpy()))


class GptjMlp(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc_in, self.fc_out = nn.Linear(128, 256), nn.Linear(256, 128)


def test_gptj_policy_quantises_fc_out_weight_only(tmp_path):
    torch.manual_seed(2)
    m = GptjMlp().half()
    cache = MixLibCache(64, device="cpu")
    done = ck.quantize_(m, 8, cache, arch="GPTJForCausalLM")
    assert done == {"fc_in": (8, False), "fc_out": (8, True)}
    assert m.fc_out.weight_only and m.fc_out.q_weight.shape == (256, 128) and not m.fc_in.weight_only
    ck.save_quantized(m, str(tmp_path), {"w_bit": 8})
    fresh = GptjMlp().half()
    ck.load_quantized(fresh, str(tmp_path), MixLibCache(64, device="cpu"), arch="GPTJForCausalLM")
    for k, v in m.state_dict().items():
        assert torch.equal(v, fresh.state_dict()[k])
    x = torch.randn(4, 256).half()
    assert torch.equal(m.fc_out(x), fresh.fc_out(x))


def _ft_preprocess_int8(qn):
    """Independent, loop-for-loop restatement of FasterTransformer's published preprocess_weights_for_mixed_gemm for int8 on
    sm75-sm90 (the routine EETQ's preprocess_weights wraps): permute_B_rows_for_mixed_gemm, subbyte_transpose,
    interleave_column_major_tensor (64 rows per tile, 2 columns interleaved), add_bias_and_interleave_int8s_inplace."""
    K, N = qn.shape
    a = np.zeros_like(qn)
    for base in range(0, K, 16):
        for tr in range(16):
            a[base + tr] = qn[base + 8 * ((tr % 4) // 2) + tr % 2 + 2 * (tr // 4)]
    tv = a.T.copy().view(np.uint32).reshape(-1)                       # column-major [N,K], 4 k-elements per word
    nvr = K // 4
    out = np.zeros_like(tv)
    for rc in range(N):
        for bvr in range(0, nvr, 16):
            for vr in range(bvr, min(nvr, bvr + 16)):
                out[(rc // 2) * nvr * 2 + 2 * bvr + 16 * (rc % 2) + vr % 16] = tv[rc * nvr + vr]
    b = (out.view(np.int8).astype(np.int16) + 128).astype(np.uint8)
    for i in range(0, b.size, 4):
        b[i + 1], b[i + 2] = b[i + 2], b[i + 1]
    return b.view(np.int8).reshape(K, N)


def test_eetq_interleave_matches_the_published_routine_and_inverts():
    torch.manual_seed(0)
    for K, N in [(64, 2), (128, 12), (256, 64)]:
        q = torch.randint(-128, 128, (K, N), dtype=torch.int8)
        p = eetq.preprocess_weights(q)
        assert p.shape == q.shape and p.dtype == torch.int8
        assert np.array_equal(p.numpy(), _ft_preprocess_int8(q.numpy()))
        assert torch.equal(eetq.unprocess_weights(p), q)
    with pytest.raises(ValueError):
        eetq.preprocess_weights(torch.zeros(100, 8, dtype=torch.int8))


def test_weight_only_checkpoint_in_the_references_layout(tmp_path):
    """A weight-only layer's q_weight is EETQ's interleaved image in a reference checkpoint and the plain matrix in ours, under the
    same key / shape / dtype: quant_config.json says which; a missing key means "written by the reference"."""
    import json
    from mixq_amd import checkpoint as ck
    from mixq_amd import MixLibCache

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc_out = torch.nn.Linear(128, 64, bias=True)
    torch.manual_seed(0)
    m = Blk().half()
    cache = MixLibCache(8, device="cpu")
    m.fc_out = MixLinear_GEMM.from_linear(m.fc_out, 8, weight_only=True, cache=cache, dev="cpu", name="fc_out")
    plain = m.fc_out.q_weight.clone()
    for layout in ("plain", "eetq"):
        d = tmp_path / layout
        ck.save_quantized(m, str(d), {"w_bit": 8}, w8a16_layout=layout)
        cfg = json.load(open(d / "quant_config.json"))
        assert cfg["w8a16_layout"] == layout
        on_disk = ck.load_state_dict_files(str(d))["fc_out.q_weight"]
        assert torch.equal(on_disk, plain if layout == "plain" else eetq.preprocess_weights(plain))
        if layout == "eetq":                                           # what the reference itself writes: no key at all
            cfg.pop("w8a16_layout")
            json.dump(cfg, open(d / "quant_config.json", "w"))
        fresh = Blk().half()
        fresh.fc_out = MixLinear_GEMM(128, 64, True, "cpu", bit=8, weight_only=True, cache=cache)
        sd = ck.load_state_dict_files(str(d))
        if cfg.get("w8a16_layout", "eetq") == "eetq":
            sd["fc_out.q_weight"] = eetq.unprocess_weights(sd["fc_out.q_weight"])
        fresh.load_state_dict(sd)
        assert torch.equal(fresh.fc_out.q_weight, plain)


def test_keyless_checkpoints_default_to_the_reference_layout(tmp_path):
    """a checkpoint without the `w8a16_layout` key is what the reference writes (interleaved image) and loads as such,
    silently.  The bytes are consulted only to recognise this library's own round-1 output (plain matrix): when EVERY weight-only
    layer reads as plain it is loaded as plain with a warning; layers that disagree, or sit near the threshold, raise instead of
    guessing (a mis-vote would load permuted weights).  An explicit argument always wins."""
    import json
    import warnings as w
    import pytest
    from mixq_amd import checkpoint as ck
    from mixq_amd import MixLibCache

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc_out = torch.nn.Linear(128, 64, bias=False)

    class Net(torch.nn.Module):
        def __init__(self, n=1):
            super().__init__()
            self.layers = torch.nn.ModuleList([Blk() for _ in range(n)])
    def gaussian(n=1):                                                 # trained weights are bell-shaped: the byte statistic is decisive
        net = Net(n)
        for b in net.layers:
            torch.nn.init.normal_(b.fc_out.weight, std=0.02)
        return net.half()
    torch.manual_seed(0)
    m = gaussian()
    cache = MixLibCache(8, device="cpu")
    ck.quantize_(m, 8, cache, arch="GPTJForCausalLM", blocks=m.layers)
    assert m.layers[0].fc_out.weight_only
    plain = m.layers[0].fc_out.q_weight.clone()
    assert ck._w8a16_stat(plain) < ck._W8A16_BAND[0] and ck._w8a16_stat(eetq.preprocess_weights(plain)) > ck._W8A16_BAND[1]
    assert ck.detect_w8a16_layout(plain) == "plain" and ck.detect_w8a16_layout(eetq.preprocess_weights(plain)) == "eetq"
    for layout in ("plain", "eetq"):
        d = tmp_path / layout
        ck.save_quantized(m, str(d), {"w_bit": 8}, w8a16_layout=layout)
        cfg = json.load(open(d / "quant_config.json"))
        assert cfg["writer"] == "mixq_amd"
        cfg.pop("w8a16_layout"); cfg.pop("writer")                   # a round-1 checkpoint of ours / a reference checkpoint
        json.dump(cfg, open(d / "quant_config.json", "w"))
        fresh = Net().half()
        if layout == "plain":
            with pytest.warns(RuntimeWarning, match="w8a16_layout"):
                ck.load_quantized(fresh, str(d), cache, arch="GPTJForCausalLM", blocks=fresh.layers)
        else:
            with w.catch_warnings():                                   # the reference's own form: the default, nothing to say
                w.simplefilter("error")
                ck.load_quantized(fresh, str(d), cache, arch="GPTJForCausalLM", blocks=fresh.layers)
        assert torch.equal(fresh.layers[0].fc_out.q_weight, plain), layout
        fresh2 = Net().half()                                          # the explicit argument wins, silently
        with w.catch_warnings():
            w.simplefilter("error")
            ck.load_quantized(fresh2, str(d), cache, arch="GPTJForCausalLM", blocks=fresh2.layers, w8a16_layout=layout)
        assert torch.equal(fresh2.layers[0].fc_out.q_weight, plain)
    # two weight-only layers that disagree (one plain, one interleaved), and one whose bytes sit between the two populations: no guess
    m2 = gaussian(2)
    ck.quantize_(m2, 8, cache, arch="GPTJForCausalLM", blocks=m2.layers)
    d = tmp_path / "mixed"
    ck.save_quantized(m2, str(d), {"w_bit": 8}, w8a16_layout="plain")
    cfg = json.load(open(d / "quant_config.json"))
    cfg.pop("w8a16_layout"); cfg.pop("writer")
    json.dump(cfg, open(d / "quant_config.json", "w"))
    sd = ck.load_state_dict_files(str(d))
    good = dict(sd)
    sd["layers.1.fc_out.q_weight"] = eetq.preprocess_weights(sd["layers.1.fc_out.q_weight"])
    torch.save(sd, str(d / "pytorch_model.bin"))
    with pytest.raises(RuntimeError, match="w8a16_layout"):
        ck.load_quantized(Net(2).half(), str(d), cache, arch="GPTJForCausalLM", blocks=None)
    amb = dict(good)                                                   # (uniform bytes - e.g. an untrained nn.Linear - read the same both ways)
    g = torch.Generator().manual_seed(1)
    amb["layers.1.fc_out.q_weight"] = torch.randint(-128, 128, good["layers.1.fc_out.q_weight"].shape, generator=g, dtype=torch.int8)   # mean |q| = 64
    torch.save(amb, str(d / "pytorch_model.bin"))
    with pytest.raises(RuntimeError, match="w8a16_layout"):
        ck.load_quantized(Net(2).half(), str(d), cache, arch="GPTJForCausalLM", blocks=None)
