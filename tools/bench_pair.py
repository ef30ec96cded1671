#!/usr/bin/env python3
"""gate_proj + up_proj of the Llama-2-7b MLP at 512 tokens (4096 -> 11008 twice, 41 outlier columns): two launches (up_proj, then gate_proj
with the MIXQ_ACT_SILU_MUL epilogue reading up's output) against ONE launch over the interleaved weight image (MIXQ_ACT_SILU_PAIR).
Same-run interleaved A/B under bench.py's protocol; both arms are checked bit-identical first.
  python tools/bench_pair.py [--tokens 512] [--shape 4096,11008] [--rounds 7] [--amax]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mixq_amd import _capi, mixlib  # noqa: E402
from mixq_amd.fused import interleave_pair_rows  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=512)
ap.add_argument("--shape", default="4096,11008")
ap.add_argument("--outliers", type=int, default=41)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--bit", type=int, default=8, choices=[8, 4], help="4: W4A4 on the FP6 pipe (give --outliers 128)")
ap.add_argument("--amax", action="store_true", help="both arms also leave down_proj's row maxima")
ap.add_argument("--cfgs", default="", help="comma-separated tilings to force for extra one-launch arms (e.g. wr128x256_s16_d3_l2)")
args = ap.parse_args()
K, N = (int(v) for v in args.shape.split(","))
M, n_out = args.tokens, args.outliers
dev = "cuda"
g = torch.Generator().manual_seed(1)
XF, WF = (1, 2) if args.bit == 8 else (_capi.FMT_R6X128, _capi.FMT_F6X128)
rnd = (lambda r, k: torch.randint(-127, 128, (r, k), generator=g, dtype=torch.int8)) if args.bit == 8 else \
      (lambda r, k: torch.randint(0, 256, (r, k // 2), generator=g, dtype=torch.uint8))        # (nibble pairs)
qx = mixlib.PackOperand(rnd(M, K).to(dev), XF)
sx = (torch.rand((M, 1), generator=g) * 0.01 + 0.001).half().to(dev)
pad = (n_out + 15) // 16 * 16
xo = (torch.randn((M, pad), generator=g) * 4).half().to(dev)[:, :n_out]
lay = {}
for nm in ("up", "gate"):
    qw = rnd(N, K).to(dev)
    sw = (torch.rand((1, N), generator=g) * 0.001 + 0.0001).half().to(dev)
    wo = (torch.randn((N, pad), generator=g) * 0.02).half().to(dev)
    lay[nm] = dict(qw=qw, wpk=mixlib.PackOperand(qw, WF), sw=sw, wo=wo[:, :n_out])
jw = mixlib.PackOperand(interleave_pair_rows(lay["up"]["qw"], lay["gate"]["qw"]), WF)
jsw = interleave_pair_rows(lay["up"]["sw"].reshape(-1), lay["gate"]["sw"].reshape(-1)).reshape(1, -1)
jwo_full = torch.zeros((2 * N, pad), dtype=torch.float16, device=dev)
jwo_full[:, :n_out] = interleave_pair_rows(lay["up"]["wo"], lay["gate"]["wo"])
jwo = jwo_full[:, :n_out]
y_up = torch.empty((M, N), dtype=torch.float16, device=dev)
y_a = torch.empty((M, N), dtype=torch.float16, device=dev)
y_b = torch.empty((M, N), dtype=torch.float16, device=dev)
amax_a = torch.zeros(M, dtype=torch.int32, device=dev)
amax_b = torch.zeros(M, dtype=torch.int32, device=dev)
mask = torch.zeros((N + 31) // 32, dtype=torch.int32, device=dev)
ex_a = {"row_amax": amax_a, "col_mask": mask} if args.amax else {}
ex_b = {"row_amax": amax_b, "col_mask": mask} if args.amax else {}


def two():
    mixlib.FusedLinear(qx, lay["up"]["wpk"], sx, lay["up"]["sw"], xo, lay["up"]["wo"], n_out, None, M, N, K, bit=args.bit, out=y_up)
    mixlib.FusedLinear(qx, lay["gate"]["wpk"], sx, lay["gate"]["sw"], xo, lay["gate"]["wo"], n_out, None, M, N, K, bit=args.bit, act=_capi.ACT_SILU_MUL,
                       addend=y_up, out=y_a, **ex_a)


def one():
    mixlib.FusedLinear(qx, jw, sx, jsw, xo, jwo, n_out, None, M, 2 * N, K, bit=args.bit, act=_capi.ACT_SILU_PAIR, out=y_b, **ex_b)


two(); one()
torch.cuda.synchronize()
assert torch.equal(y_a, y_b), f"{int((y_a != y_b).sum())} elements differ"
if args.amax:
    assert torch.equal(amax_a, amax_b)
st = torch.cuda.Stream()
graphs = {}
lib = _capi.load()
names = _capi.gemm_config_names()
arms = [("two launches (up, gate x up)", two, -1), ("one launch (interleaved rows)", one, -1)]
arms += [(f"one launch, {c}", one, names.index(c)) for c in args.cfgs.split(",") if c]
for name, fn, cfg in arms:
    assert lib.mixq_gemm_set_config(cfg) == 0
    with torch.cuda.stream(st):
        fn()
        if cfg >= 0:
            torch.cuda.synchronize()
            assert torch.equal(y_a, y_b), name
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(args.steps):
                fn()
    graphs[name] = gr
lib.mixq_gemm_set_config(-1)
res = {k: [] for k in graphs}
first = {}
with torch.cuda.stream(st):
    for r in range(args.rounds):
        for name, gr in graphs.items():
            ms, f, _ = bench.conditioned_replay(gr, st)
            res[name].append(ms * 1e3 / args.steps)
            first.setdefault(name, f * 1e3 / args.steps)
print(f"gate_proj + up_proj, W{args.bit}A{args.bit}, {M} tokens, {K} -> {N} (x2), {n_out} outlier columns{', with row maxima' if args.amax else ''}; "
      f"{args.steps} steps per graph, {args.rounds} interleaved rounds; outputs bit-identical")
for name, v in res.items():
    print(f"  {name:44s} median {statistics.median(v):7.2f} us   min {min(v):7.2f}   (first replay {first[name]:.2f})   all: " + " ".join(f"{x:.2f}" for x in v))
