#!/usr/bin/env python3
"""A handful of launches of the shipped fused GEMM at the metric shape (for rocprofv3 --pmc passes: counters serialise every kernel, so the
profiled command must be SHORT).   python tools/launch_few.py [--bit 4|8] [--n 40] [--nout 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import mixlib  # noqa: E402
from mixq_amd.linear import pack_to_i4  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="512x11008x4096")
ap.add_argument("--bit", type=int, default=4)
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--nout", type=int, default=-1)
args = ap.parse_args()
M, N, K = (int(v) for v in args.shape.split("x"))
nout = args.nout if args.nout >= 0 else (128 if args.bit == 4 else 41)
dev = "cuda"
g = torch.Generator().manual_seed(0)
if args.bit == 4:
    wf = mixlib.PackOperand(pack_to_i4(torch.randint(-8, 8, (N, K), generator=g, dtype=torch.int8)).to(dev), 3)
    xp = mixlib.PackOperand(pack_to_i4(torch.randint(-7, 8, (M, K), generator=g, dtype=torch.int8)).to(dev), 4)
else:
    wf = mixlib.PackOperand(torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev), 2)
    xp = mixlib.PackOperand(torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(dev), 1)
sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
pad = (max(nout, 1) + 15) // 16 * 16
xo = torch.randn((M, pad), device=dev).half()[:, :nout] if nout else None
wo = torch.randn((N, pad), device=dev).half()[:, :nout] if nout else None
out = torch.empty(M, N, dtype=torch.float16, device=dev)
for _ in range(args.n):
    mixlib.FusedLinear(xp, wf, sx, sw, xo, wo, nout, None, M, N, K, bit=args.bit, out=out)
torch.cuda.synchronize()
print("done", args.n)
