"""checkpoint format, layer policy, loader and QKV fusion (host logic, CPU).

The reference's layout is pinned by tests/golden/g6_checkpoint_layout.json (state_dict keys / shapes / dtypes of the
reference's own module per flavour and its policy tables, captured by oracle/gen_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from mixq_amd import MixLibCache, MixLinear_GEMM
from mixq_amd import checkpoint as ck

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g6():
    with open(os.path.join(GOLD, "g6_checkpoint_layout.json")) as f:
        return json.load(f)


def layout(m):
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}


def test_policy_tables_match_reference(g6):
    assert list(ck.EIGHTBIT_ONLY) == g6["eightbit_only_name"]
    assert {k: list(v) for k, v in ck.WEIGHT_ONLY.items()} == g6["weight_only_map"]


def test_state_dict_layout_matches_reference(g6):
    cache8, cache4 = MixLibCache(64, device="cpu"), MixLibCache(64, bit=4, device="cpu")
    l8 = nn.Linear(256, 96, bias=True).half()
    q8 = MixLinear_GEMM.from_linear(l8, bit=8, cache=cache8, dev="cpu", init_only=True)
    assert layout(q8) == g6["w8_bias"]
    l4 = nn.Linear(512, 64, bias=False).half()
    q4 = MixLinear_GEMM.from_linear(l4, bit=4, cache=cache4, dev="cpu", init_only=True)
    assert layout(q4) == g6["w4_nobias"]
    wo = MixLinear_GEMM.from_linear(l8, bit=8, weight_only=True, cache=cache8, dev="cpu", init_only=True)
    assert layout(wo) == g6["weight_only_w8_bias"]


def test_reference_tensors_load_strict(golden):
    """A state_dict with the reference's tensors (G2 / G3) loads with strict=True and reads back bit-identically."""
    g2, g3 = golden("g2_from_linear_w8.npz"), golden("g3_from_linear_w4.npz")
    q8 = MixLinear_GEMM(256, 96, True, "cpu", bit=8, cache=MixLibCache(64, device="cpu"))
    sd = {"q_weight": torch.from_numpy(g2["q_weight"]), "scale_col": torch.from_numpy(g2["scale_col"]), "bias": torch.from_numpy(g2["bias"])}
    q8.load_state_dict(sd, strict=True)
    assert np.array_equal(q8.q_weight.numpy(), g2["q_weight"]) and np.array_equal(q8.scale_col.numpy(), g2["scale_col"])
    q4 = MixLinear_GEMM(512, 64, False, "cpu", bit=4, cache=MixLibCache(64, bit=4, device="cpu"))
    sd = {k: torch.from_numpy(g3[k]) for k in ("q_weight", "scale_col", "weight_cache", "ind")}
    q4.load_state_dict(sd, strict=True)
    for k in sd:
        assert np.array_equal(getattr(q4, k).numpy(), g3[k])


@pytest.mark.parametrize("name,w_bit,arch,expect", [
    ("self_attn.q_proj", 8, "LlamaForCausalLM", (8, False)),
    ("self_attn.q_proj", 4, "LlamaForCausalLM", (4, False)),
    ("self_attn.o_proj", 4, "LlamaForCausalLM", (8, False)),
    ("mlp.down_proj", 4, "MistralForCausalLM", (8, False)),
    ("mlp.up_proj", 4, "LlamaForCausalLM", (4, False)),
    ("mlp.fc_out", 8, "GPTJForCausalLM", (8, True)),
    ("mlp.fc_out", 4, "GPTJForCausalLM", (8, False)),        # eight-bit-only overrides weight-only (mixquant.py:190-195)
    ("mlp.fc_in", 8, "GPTJForCausalLM", (8, False)),
])
def test_layer_policy(name, w_bit, arch, expect):
    assert ck.layer_policy(name, w_bit, arch) == expect


def test_layer_policy_extra_weight_only():
    assert ck.layer_policy("mlp.down_proj", 8, "LlamaForCausalLM", "down_proj,foo") == (8, True)
    assert ck.layer_policy("mlp.up_proj", 8, "LlamaForCausalLM", "down_proj,foo") == (8, False)


class Attn(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (nn.Linear(h, h, bias=False) for _ in range(4))


class Mlp(nn.Module):
    def __init__(self, h, f):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = nn.Linear(h, f, bias=False), nn.Linear(h, f, bias=False), nn.Linear(f, h, bias=False)


class Block(nn.Module):
    def __init__(self, h, f):
        super().__init__()
        self.self_attn, self.mlp, self.norm = Attn(h), Mlp(h, f), nn.LayerNorm(h)


class Tiny(nn.Module):
    def __init__(self, h=128, f=256, n=2):
        super().__init__()
        self.embed = nn.Embedding(32, h)
        self.layers = nn.ModuleList([Block(h, f) for _ in range(n)])
        self.lm_head = nn.Linear(h, 32, bias=False)


def make_scales(model, h, f):
    g = torch.Generator().manual_seed(11)
    sc = {}
    for i, _ in enumerate(model.layers):
        per_input = {}                                  # q/k/v (and gate/up) see the same input -> the same activation scales
        for name, lin in ck.named_linears(model.layers[i]).items():
            src = "attn_in" if name.split(".")[-1] in ("q_proj", "k_proj", "v_proj") else \
                  "mlp_in" if name.split(".")[-1] in ("gate_proj", "up_proj") else name
            if src not in per_input:
                per_input[src] = torch.rand(lin.in_features, generator=g) * 4 + 0.1
            sc[f"model.layers.{i}.{name}"] = per_input[src]
    return sc


@pytest.mark.parametrize("w_bit,safetensors,shard", [(8, False, "10GB"), (4, False, "40KB"), (8, True, "60KB"), (4, True, "10GB")])
def test_quantize_save_load_roundtrip(tmp_path, w_bit, safetensors, shard):
    torch.manual_seed(0)
    model = Tiny().half()
    cache = MixLibCache(64, bit=w_bit, device="cpu")
    scales = make_scales(model, 128, 256) if w_bit == 4 else None
    done = ck.quantize_(model, w_bit, cache, blocks=model.layers, act_scales=scales)
    assert done["0.self_attn.q_proj"] == (w_bit, False) and done["1.mlp.down_proj"] == (8, False) and done["0.self_attn.o_proj"] == (8, False)
    assert isinstance(model.lm_head, nn.Linear)                       # only decoder blocks are quantised
    assert isinstance(model.layers[0].norm, nn.LayerNorm)
    files = ck.save_quantized(model, str(tmp_path), {"w_bit": w_bit, "zero_point": True}, safetensors=safetensors, shard_size=shard)
    qc = json.load(open(tmp_path / "quant_config.json"))
    assert qc["w_bit"] == w_bit and qc["version"] == "MIX"
    ext = ".safetensors" if safetensors else ".bin"
    assert all(f.endswith(ext) for f in files)
    if len(files) > 1:
        idx = json.load(open(tmp_path / (("model.safetensors" if safetensors else "pytorch_model.bin") + ".index.json")))
        assert set(idx["weight_map"]) == set(model.state_dict()) and set(idx["weight_map"].values()) == set(files)
        assert idx["metadata"]["total_size"] == sum(v.numel() * v.element_size() for v in model.state_dict().values())
        assert files[0].endswith(f"-00001-of-{len(files):05d}{ext}")
    else:
        assert files == [("model.safetensors" if safetensors else "pytorch_model.bin")]

    keys = set(model.state_dict())
    assert "layers.0.self_attn.q_proj.q_weight" in keys and "layers.0.self_attn.q_proj.scale_col" in keys
    assert ("layers.0.self_attn.q_proj.ind" in keys) == (w_bit == 4)
    assert "layers.0.self_attn.o_proj.ind" not in keys               # 8-bit: runtime state, not in the checkpoint

    fresh = Tiny().half()
    cache2 = MixLibCache(64, bit=w_bit, device="cpu")
    qc2 = ck.load_quantized(fresh, str(tmp_path), cache2, blocks=fresh.layers)
    assert qc2 == qc
    a, b = model.state_dict(), fresh.state_dict()
    assert set(a) == set(b)
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
    q = fresh.layers[1].self_attn.q_proj
    assert isinstance(q, MixLinear_GEMM) and q.bit == w_bit and q.cache is cache2


def test_load_rejects_other_versions(tmp_path):
    json.dump({"w_bit": 4, "version": "QUIK"}, open(tmp_path / "quant_config.json", "w"))
    with pytest.raises(NotImplementedError):
        ck.load_quantized(Tiny().half(), str(tmp_path), MixLibCache(64, device="cpu"))


def test_read_quant_config_default(tmp_path):
    assert ck.read_quant_config(str(tmp_path)) == {"w_bit": 0, "version": "MIX"}       # base.py:246-247


def test_shard_size_parse():
    assert ck._parse_size("10GB") == 10 * 10 ** 9 and ck._parse_size("512MiB") == 512 * 2 ** 20 and ck._parse_size(123) == 123
    with pytest.raises(ValueError):
        ck._parse_size("ten")


@pytest.mark.parametrize("w_bit", [8, 4])
def test_fuse_qkv(w_bit):
    torch.manual_seed(1)
    h = 256
    attn = Attn(h).half()
    cache = MixLibCache(64, bit=w_bit, device="cpu")
    ls = torch.rand(h, generator=torch.Generator().manual_seed(2)) + 0.1
    qs = [MixLinear_GEMM.from_linear(l, bit=w_bit, cache=cache, dev="cpu", layer_scales=ls if w_bit == 4 else None)
          for l in (attn.q_proj, attn.k_proj, attn.v_proj)]
    fused = ck.fuse_qkv(*qs, cache)
    assert fused.out_features == 3 * h and fused.in_features == h and fused.bias is None and fused.bit == w_bit
    assert torch.equal(fused.q_weight, torch.cat([q.q_weight for q in qs], 0))
    assert torch.equal(fused.scale_col, torch.cat([q.scale_col for q in qs], 1)) and fused.scale_col.shape == (1, 3 * h)
    if w_bit == 4:
        assert torch.equal(fused.ind, qs[0].ind)
        assert torch.equal(fused.weight_cache, torch.cat([q.weight_cache for q in qs], 0))
    else:
        assert fused.ind.numel() == 0 and fused.weight_cache is None


def test_fuse_qkv_rejects_mismatch():
    cache = MixLibCache(64, bit=4, device="cpu")
    h = 256
    a = Attn(h).half()
    g = torch.Generator().manual_seed(3)
    q = MixLinear_GEMM.from_linear(a.q_proj, bit=4, cache=cache, dev="cpu", layer_scales=torch.rand(h, generator=g))
    k = MixLinear_GEMM.from_linear(a.k_proj, bit=4, cache=cache, dev="cpu", layer_scales=torch.rand(h, generator=g))
    with pytest.raises(ValueError):
        ck.fuse_qkv(q, k, k, cache)
    with pytest.raises(TypeError):
        ck.fuse_qkv(q, a.k_proj, k, cache)
