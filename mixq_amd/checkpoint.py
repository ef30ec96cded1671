"""Checkpoint format and layer-replacement policy of MixQ, so checkpoints written by the reference
load into `mixq_amd.MixLinear_GEMM` and vice versa.

On-disk contract (reference `mixquant/models/base.py:78-119`, `:161-229`, `:273-347`):
  * `quant_config.json`  {"w_bit": 8 | 4, "version": "MIX", ...}                                  (base.py:117-119, :240-247)
  * weight shards in the Hugging Face layout: `pytorch_model.bin` (torch.save) or `model.safetensors`; more than one
    shard -> `pytorch_model-0000i-of-0000n.bin` + `<name>.index.json` {"metadata": {"total_size"}, "weight_map"}  (:97-114)
  * per quantised Linear `<prefix>`: `<prefix>.q_weight`, `<prefix>.scale_col`, `<prefix>.bias` when present, and for the
    4-bit layers `<prefix>.weight_cache` + `<prefix>.ind` (registered buffers, modules/linear.py:41-78).  The 8-bit layers'
    `ind` / `weight_cache` are runtime state discovered online and are NOT part of the checkpoint.
Policy (reference `utils/module.py:2-12`, applied in `quantize/mixquant.py:163-262` and `base.py:289-330`):
  * with w_bit = 4, layers whose name contains down_proj / o_proj / fc_out stay 8-bit,
  * per architecture, some layers are weight-only W8A16 (GPT-J `fc_out`); an optional comma list adds more.

Weight-only (W8A16) layers are the one place where the two sides store DIFFERENT bytes under the same key, shape and
dtype: the reference's `q_weight` there is EETQ's CUTLASS-interleaved image (modules/linear.py:102-106), ours the plain
int8 [K,N] matrix.  `quant_config.json` therefore records `"w8a16_layout"`: "plain" (written by save_quantized) or
"eetq"; load_quantized converts "eetq" layers with mixq_amd.eetq.unprocess_weights, and save_quantized(...,
w8a16_layout="eetq") writes the reference's form.  A checkpoint WITHOUT the key is the reference's own (it writes none: interleaved image) and is
loaded as "eetq".  The one other keyless writer was this library's round 1 (plain matrix, no `writer` marker yet): the bytes tell
the two apart decisively (`detect_w8a16_layout`: the interleaved image is offset-binary, q + 128, so read as int8 its values crowd
the two ends of the range - mean |q| ~ 100 - whereas quantised weights crowd zero - ~ 20-45), so a keyless checkpoint whose
weight-only layers ALL read as plain is loaded as such with a warning, and one whose layers disagree or sit near the threshold
raises - a silent mis-vote would load permuted weights.  save_quantized also records `"writer": "mixq_amd"`.  (EETQ is un-versioned and
absent from the reference tree: the interleave is restated from the FasterTransformer routine it wraps - parity for these
layers stays UNPINNED, see mixq_amd/eetq.py.)

Nothing here touches the GPU by itself: quantisation uses torch ops on whatever device the layer lives on; the kernels
are only reached through `MixLinear_GEMM.forward`.
"""
import json
import os
import re
import warnings

import torch
import torch.nn as nn

from .linear import MixLinear_GEMM

EIGHTBIT_ONLY = ("down_proj", "o_proj", "fc_out")                      # utils/module.py:2
WEIGHT_ONLY = {                                                         # utils/module.py:4-12
    "GPTJForCausalLM": ("fc_out",),
    "LlamaForCausalLM": (),
    "AquilaForCausalLM": (),
    "BaichuanForCausalLM": (),
    "MistralForCausalLM": (),
    "FalconForCausalLM": (),
}
QUANT_CONFIG = "quant_config.json"


def layer_policy(name, w_bit, arch="LlamaForCausalLM", extra_weight_only=""):
    """(bit, weight_only) for the Linear called `name` inside a decoder block (mixquant.py:181-212, base.py:296-313)."""
    weight_only = any(key in name for key in WEIGHT_ONLY.get(arch, ()))
    bit = w_bit
    if bit == 4 and any(key in name for key in EIGHTBIT_ONLY):
        bit, weight_only = 8, False
    if extra_weight_only:
        if any(key and key in name for key in extra_weight_only.split(",")):
            weight_only = True
    return bit, weight_only


def named_linears(module):
    """{relative name: nn.Linear} of every plain Linear under `module` (utils/module.py:14-15)."""
    return {name: m for name, m in module.named_modules() if isinstance(m, nn.Linear)}


def set_by_name(root, name, new_module):
    """Replace the sub-module `name` (dotted, digits index containers) of `root` (utils/module.py:25-36)."""
    parts = name.split(".")
    mod = root
    for p in parts[:-1]:
        mod = mod[int(p)] if p.isdigit() else getattr(mod, p)
    if parts[-1].isdigit():
        mod[int(parts[-1])] = new_module
    else:
        setattr(mod, parts[-1], new_module)


def _blocks(root, blocks):
    if blocks is None:
        return [("", root)]
    return [(str(i), b) for i, b in enumerate(blocks)]


@torch.no_grad()
def quantize_(root, w_bit, cache, arch="LlamaForCausalLM", blocks=None, act_scales=None, scale_key=None,
              extra_weight_only="", dev=None):
    """Replace every nn.Linear under `root` (or under each decoder block of `blocks`) by a quantised MixLinear_GEMM
    (`quantize/mixquant.py:163-262`).  4-bit layers need per-input-channel activation scales: `act_scales[scale_key(block
    index, name)]` (the reference reads `act_scales/<model>.pt` keyed `model.layers.<i>.<name>`, mixquant.py:197-206).
    Returns {qualified name: (bit, weight_only)}."""
    done = {}
    for bidx, block in _blocks(root, blocks):
        for name, lin in named_linears(block).items():
            bit, weight_only = layer_policy(name, w_bit, arch, extra_weight_only)
            layer_scales = None
            if bit == 4:
                if act_scales is None:
                    raise ValueError("4-bit MixQ layers need activation scales (quantize/mixquant.py:197-206)")
                key = scale_key(bidx, name) if scale_key else (f"model.layers.{bidx}.{name}" if bidx != "" else name)
                layer_scales = act_scales[key]
            d = dev if dev is not None else lin.weight.device
            q = MixLinear_GEMM.from_linear(lin, bit=bit, weight_only=weight_only, init_only=False, cache=cache,
                                           layer_scales=layer_scales, dev=d, name=bidx + name)
            set_by_name(block, name, q)
            done[(bidx + "." if bidx != "" else "") + name] = (bit, weight_only)
    return done


def prepare_(root, quant_config, cache, arch="LlamaForCausalLM", blocks=None, dev=None):
    """Replace every nn.Linear by an EMPTY MixLinear_GEMM of the checkpoint's geometry (`init_only=True`), ready for
    `load_state_dict` (`base.py:273-347` `_load_mix_quantized_modules`)."""
    if quant_config.get("version", "MIX").upper() != "MIX":
        raise NotImplementedError(f"quant_config version {quant_config.get('version')!r}: only MIX checkpoints are on this path")
    w_bit = int(quant_config["w_bit"])
    done = {}
    for bidx, block in _blocks(root, blocks):
        for name, lin in named_linears(block).items():
            bit, weight_only = layer_policy(name, w_bit, arch)
            d = torch.device(dev if dev is not None else lin.weight.device)
            if d.type == "meta":                       # skeleton built under init_empty_weights()
                d = torch.device("cpu")
            q = MixLinear_GEMM.from_linear(lin, bit=bit, weight_only=weight_only, init_only=True, cache=cache, dev=d,
                                           name=bidx + name, fp_features_num=128)
            set_by_name(block, name, q)
            done[(bidx + "." if bidx != "" else "") + name] = (bit, weight_only)
    return done


# ---- shards ------------------------------------------------------------------------------------------------------
_UNITS = {"KB": 10 ** 3, "MB": 10 ** 6, "GB": 10 ** 9, "KIB": 2 ** 10, "MIB": 2 ** 20, "GIB": 2 ** 30}


def _parse_size(s):
    if isinstance(s, int):
        return s
    m = re.fullmatch(r"\s*(\d+(?:\.\d+)?)\s*([A-Za-z]+)\s*", s)
    if not m or m.group(2).upper() not in _UNITS:
        raise ValueError(f"shard size {s!r}: expected e.g. '10GB'")
    return int(float(m.group(1)) * _UNITS[m.group(2).upper()])


def shard_state_dict(state_dict, max_shard_size="10GB", weights_name="pytorch_model.bin"):
    """Greedy split in key order, Hugging Face file naming.  Returns ({file: {key: tensor}}, index or None)."""
    limit = _parse_size(max_shard_size)
    shards, cur, cur_bytes, total = [], {}, 0, 0
    for k, v in state_dict.items():
        nbytes = v.numel() * v.element_size()
        if cur and cur_bytes + nbytes > limit:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[k] = v
        cur_bytes += nbytes
        total += nbytes
    shards.append(cur)
    if len(shards) == 1:
        return {weights_name: shards[0]}, None
    stem, ext = os.path.splitext(weights_name)
    files, weight_map = {}, {}
    for i, sh in enumerate(shards):
        fn = f"{stem}-{i + 1:05d}-of-{len(shards):05d}{ext}"
        files[fn] = sh
        for k in sh:
            weight_map[k] = fn
    return files, {"metadata": {"total_size": total}, "weight_map": weight_map}


def _weight_only_prefixes(root):
    return [name + "." if name else "" for name, m in root.named_modules() if isinstance(m, MixLinear_GEMM) and m.weight_only]


def save_quantized(root, save_dir, quant_config, safetensors=False, shard_size="10GB", w8a16_layout="plain"):
    """Write `root.state_dict()` + `quant_config.json` in the reference's layout (`base.py:78-119`).  When `root` is a
    Hugging Face model its config files are written too (as the reference's `save_pretrained(state_dict={})` does).
    `w8a16_layout`: how weight-only layers' q_weight is stored - "plain" (ours) or "eetq" (the reference's interleaved image)."""
    if w8a16_layout not in ("plain", "eetq"):
        raise ValueError("w8a16_layout must be 'plain' or 'eetq'")
    os.makedirs(save_dir, exist_ok=True)
    cfg = dict(quant_config)
    cfg.setdefault("version", "MIX")
    cfg["w8a16_layout"] = w8a16_layout
    cfg["writer"] = "mixq_amd"
    if hasattr(root, "save_pretrained") and hasattr(root, "config"):
        root.config.save_pretrained(save_dir)
    weights_name = "model.safetensors" if safetensors else "pytorch_model.bin"
    sd = {k: v.detach().cpu() for k, v in root.state_dict().items()}
    if w8a16_layout == "eetq":
        from . import eetq
        for pre in _weight_only_prefixes(root):
            sd[pre + "q_weight"] = eetq.preprocess_weights(sd[pre + "q_weight"])
    files, index = shard_state_dict(sd, shard_size, weights_name)
    for fn, shard in files.items():
        path = os.path.join(save_dir, fn)
        if safetensors:
            from safetensors.torch import save_file
            save_file({k: v.clone().contiguous() for k, v in shard.items()}, path, metadata={"format": "pt"})
        else:
            torch.save(shard, path)
    if index is not None:
        with open(os.path.join(save_dir, weights_name + ".index.json"), "w") as f:
            json.dump(index, f, indent=4)
    with open(os.path.join(save_dir, QUANT_CONFIG), "w") as f:
        json.dump(cfg, f, indent=4)
    return sorted(files)


def read_quant_config(load_dir, version="MIX"):
    """`quant_config.json`, or the reference's "online quantisation" default when absent (base.py:240-247)."""
    p = os.path.join(load_dir, QUANT_CONFIG)
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return {"w_bit": 0, "version": version}


def _shard_files(load_dir):
    for weights_name in ("model.safetensors", "pytorch_model.bin"):
        idx = os.path.join(load_dir, weights_name + ".index.json")
        if os.path.exists(idx):
            with open(idx) as f:
                return sorted(set(json.load(f)["weight_map"].values()))
        if os.path.exists(os.path.join(load_dir, weights_name)):
            return [weights_name]
    raise FileNotFoundError(f"no pytorch_model.bin / model.safetensors (or index) under {load_dir}")


def load_state_dict_files(load_dir):
    sd = {}
    for fn in _shard_files(load_dir):
        path = os.path.join(load_dir, fn)
        if fn.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd.update(load_file(path))
        else:
            sd.update(torch.load(path, map_location="cpu", weights_only=True))
    return sd


def _w8a16_stat(q_weight):
    return q_weight.detach().to("cpu").view(torch.int8).to(torch.int16).abs().float().mean().item()


def detect_w8a16_layout(q_weight):
    """"eetq" or "plain" from the bytes of a weight-only layer's q_weight (int8 [K,N]).  EETQ's image stores q + 128: viewed as int8
    a typical weight (|q| small) lands next to -128 / 127, so the mean magnitude is ~100; a plain per-column symmetric quantisation
    has its mass around zero (mean |q| ~ 20-45 for Gaussian-like weights scaled to absmax 127).  Threshold 64; `_W8A16_BAND` is the
    margin inside which load_quantized refuses to decide."""
    return "eetq" if _w8a16_stat(q_weight) > 64.0 else "plain"


_W8A16_BAND = (52.0, 80.0)


def load_quantized(root, load_dir, cache, arch="LlamaForCausalLM", blocks=None, dev=None, strict=True, w8a16_layout=None):
    """`from_quantized` for an already-constructed skeleton (`base.py:161-229`): swap the Linears for empty quantised
    layers per `quant_config.json`, then load the shards.  Returns the quant config.
    `w8a16_layout` ("plain" | "eetq") overrides what quant_config.json says about weight-only layers; when neither says anything the
    checkpoint is the reference's ("eetq") unless every weight-only layer's bytes say plain (module docstring)."""
    quant_config = read_quant_config(load_dir)
    prepare_(root, quant_config, cache, arch, blocks, dev)
    sd = load_state_dict_files(load_dir)
    layout = w8a16_layout if w8a16_layout is not None else quant_config.get("w8a16_layout")
    wo_keys = [pre + "q_weight" for pre in _weight_only_prefixes(root) if pre + "q_weight" in sd]
    if layout is None and wo_keys:
        # No key: what the reference writes (interleaved image) - the default.  The bytes are consulted only to recognise this library's
        # own round-1 files (plain matrix, written before the `writer` marker existed); anything in between is an error, not a guess.
        stats = [_w8a16_stat(sd[k]) for k in wo_keys]
        near = [k for k, v in zip(wo_keys, stats) if _W8A16_BAND[0] <= v <= _W8A16_BAND[1]]
        votes = ["eetq" if v > 64.0 else "plain" for v in stats]
        if near or len(set(votes)) > 1:
            raise RuntimeError(f"{load_dir}: quant_config.json has no 'w8a16_layout' key and the {len(wo_keys)} weight-only layer(s) do not "
                               f"agree on a layout by their bytes ({votes.count('eetq')} look EETQ-interleaved, {votes.count('plain')} plain, "
                               f"{len(near)} undecidable, e.g. {(near or wo_keys)[0]}): pass w8a16_layout='eetq' (a reference checkpoint) or "
                               f"'plain' to load_quantized")
        layout = votes[0]
        if layout == "plain":
            warnings.warn(f"{load_dir}: quant_config.json has no 'w8a16_layout' key; a keyless checkpoint is normally the reference's "
                          f"(EETQ-interleaved), but all {len(wo_keys)} weight-only layer(s) hold plain [K,N] int8 by their bytes (a "
                          f"round-1 checkpoint of this library) and are loaded as such - pass w8a16_layout= to decide yourself",
                          RuntimeWarning, stacklevel=2)
    if layout is None:
        layout = "plain"                                     # no weight-only layers: nothing to convert
    if layout not in ("plain", "eetq"):
        raise ValueError(f"quant_config.json: unknown w8a16_layout {layout!r}")
    if layout == "eetq":
        from . import eetq
        for pre in _weight_only_prefixes(root):
            if pre + "q_weight" in sd:
                sd[pre + "q_weight"] = eetq.unprocess_weights(sd[pre + "q_weight"])
    root.load_state_dict(sd, strict=strict)
    return quant_config


# ---- QKV fusion ----------------------------------------------------------------------------------------------------
@torch.no_grad()
def fuse_qkv(q_proj, k_proj, v_proj, cache, w_bit=None):
    """One MixLinear_GEMM computing [q; k; v] (`models/llama.py:98-157`): weights and scales concatenated along N; a
    4-bit layer concatenates `weight_cache` and takes `ind` from q_proj (the three share their input, so the static
    outlier columns chosen from the input's activation scales are the same)."""
    for m in (q_proj, k_proj, v_proj):
        if not isinstance(m, MixLinear_GEMM):
            raise TypeError("fuse_qkv: all three projections must be MixLinear_GEMM")
    if q_proj.bias is not None:
        raise NotImplementedError("fuse_qkv with bias (the reference raises here too, llama.py:146-147)")
    bit = q_proj.bit if w_bit is None else w_bit
    if not (q_proj.bit == k_proj.bit == v_proj.bit == bit):
        raise ValueError("fuse_qkv: projections disagree on the bit width")
    if q_proj.weight_only or k_proj.weight_only or v_proj.weight_only:
        raise NotImplementedError("fuse_qkv of weight-only layers")
    dev = q_proj.q_weight.device
    N = q_proj.out_features + k_proj.out_features + v_proj.out_features
    fused = MixLinear_GEMM(q_proj.in_features, N, False, dev, bit=bit, weight_only=False, cache=cache,
                           fp_features_num=getattr(q_proj, "fp_features_num", 128))
    fused.q_weight.copy_(torch.cat([q_proj.q_weight, k_proj.q_weight, v_proj.q_weight], dim=0))
    fused.scale_col.copy_(torch.cat([q_proj.scale_col, k_proj.scale_col, v_proj.scale_col], dim=1))
    if bit == 4:
        if not (torch.equal(q_proj.ind, k_proj.ind) and torch.equal(q_proj.ind, v_proj.ind)):
            raise ValueError("fuse_qkv: 4-bit projections must share their static outlier columns")
        fused.weight_cache.copy_(torch.cat([q_proj.weight_cache, k_proj.weight_cache, v_proj.weight_cache], dim=0))
        fused.ind.copy_(q_proj.ind)
    return fused
