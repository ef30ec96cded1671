#!/usr/bin/env python3
"""The Linear path of ONE Llama-2-7b decoder layer chained as the reference wires it (mixquant/models/llama.py:20-22,
modules/fused/attn.py:219,263, modules/fused/mlp.py:57-70), at 512 tokens:

    input_layernorm (+ extract + quantise for W_pack) -> W_pack 4096 -> 12288 (unfused=False) -> [attention: out of scope, its output is
    taken to be the first 4096 columns of the projection] -> o_proj 4096 -> 4096 (unfused=True) -> post_attention_layernorm (+ quantise
    for up_proj) -> MixLlamaMLP (up_proj, gate_proj with SiLU * up in the epilogue, down_proj with the row maxima from gate's epilogue)

over `--layers` independent copies of the weights (32 x 202 MB = 6.5 GB for the 7b shapes): launch i of the graph runs layer i % layers, so
every layer's weights come from HBM, as in a model - and, for comparison, over ONE copy (everything resident in the 256 MB memory-side cache
after the first pass, as in a single-layer benchmark loop).  Measurement protocol: bench.conditioned_replay.  A per-kernel breakdown comes
from running this script under `rocprofv3 --kernel-trace --stats` (round 4's evidence script did, into profiles/r04_block_kernels.txt)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import FasterTransformerRMSNorm, MixLibCache, MixLinear_GEMM, MixLlamaMLP, fused, mixlib  # noqa: E402
import bench  # noqa: E402


PROBE = ""
_sink = None


class Layer:
    def __init__(self, H, F, QKV, cache, dev, bit, seed):
        torch.manual_seed(seed)
        g = torch.Generator().manual_seed(100 + seed)
        scales = torch.rand(H, generator=g) + 0.1

        def mk(k, n, b=bit, with_scales=True):
            lin = torch.nn.Linear(k, n, bias=False).half()
            if b == 4:
                return MixLinear_GEMM.from_linear(lin, 4, cache=cache, dev=dev, layer_scales=(scales if k == H else torch.rand(k, generator=g) + 0.1))
            return MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=dev)
        self.W_pack, self.o_proj = mk(H, QKV), mk(H, H, 8)                      # (o_proj / down_proj stay 8-bit: utils/module.py:2)
        gate, up, down = mk(H, F), mk(H, F), mk(F, H, 8)
        self.mlp = MixLlamaMLP(gate, down, up, cache)
        self.norm1 = FasterTransformerRMSNorm((torch.rand(H) + 0.5).half().to(dev), 1e-5, cache)
        self.norm2 = FasterTransformerRMSNorm((torch.rand(H) + 0.5).half().to(dev), 1e-5, cache)
        self.norm1.next_layer, self.norm2.next_layer = self.W_pack, up
        self.H = H

    def __call__(self, x):
        h = self.norm1(x)
        qkv = self.W_pack(h)                                                    # (cache filled by the norm: unfused=False)
        attn = qkv[:, :self.H].contiguous()                                     # stand-in for the attention output (out of scope)
        o = self.o_proj(attn, None, True)
        h2 = self.norm2(o)
        return self.mlp(h2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=512)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--bit", type=int, default=8)
    ap.add_argument("--passes", type=int, default=2, help="passes over all layers per graph")
    args = ap.parse_args()
    dev = "cuda"
    M, H, F, QKV = args.M, 4096, 11008, 12288
    cache = MixLibCache(M, sigma=6, bit=args.bit, device=dev)
    g = torch.Generator().manual_seed(1)
    cols = torch.randperm(H, generator=g)[:41]
    base = torch.randn(M, H, generator=g).half()
    base[:, cols] *= 20
    base = base.to(dev)
    flops = 2.0 * M * (H * QKV + H * H + 2 * H * F + F * H)
    wbytes = (H * QKV + H * H + 2 * H * F + F * H) * (1 if args.bit == 8 else 0.75)
    print(f"Llama-2-7b decoder layer, Linear path only (norm, W_pack, o_proj, norm, MLP), {M} tokens, W{args.bit}A{args.bit}; {flops / 1e9:.1f} GFLOP and "
          f"~{wbytes / 1e6:.0f} MB of weights per layer; protocol: bench.conditioned_replay")
    # (a third arm - the next GEMM's weight image prefetched into the memory-side cache from a side stream - was measured in round 4 and
    # removed with mixq_prefetch in round 5: profiles/r04_prefetch_side_stream.txt, a cross-stream edge in a captured graph costs ~14 us)
    arms = [(1, "ONE layer copy (weights resident in the memory-side cache: a benchmark loop)"),
            (args.layers, f"{args.layers} layer copies in rotation (weights from HBM: a model)")]
    for nl, label in arms:
        layers = [Layer(H, F, QKV, cache, dev, args.bit, s) for s in range(nl)]
        for ly in layers:
            for _ in range(3):                                                  # outlier prediction warm-up (host syncs allowed here)
                ly(base.clone())
        torch.cuda.synchronize()
        steps = max(nl * args.passes, 16)
        xs = base.unsqueeze(0).repeat(steps, 1, 1).contiguous()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                for i in range(steps):
                    layers[i % nl](xs[i])
            torch.cuda.synchronize()
            ms, first_ms, reps = bench.conditioned_replay(gr, side, restore=lambda: xs.copy_(base.unsqueeze(0).expand_as(xs)))
        us = ms * 1e3 / steps
        print(f"  {label}: {us:7.1f} us per layer  ({flops / us / 1e6:6.0f} effective TFLOPS, {100 * flops / us / 1e6 / 5033:4.1f} % of 5033; "
              f"first replay {first_ms * 1e3 / steps:7.1f} us; {reps} untimed replays)", flush=True)
        del layers, gr, xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
