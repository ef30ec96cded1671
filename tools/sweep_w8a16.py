#!/usr/bin/env python3
"""Time the weight-only W8A16 GEMM (graph replay) per config and shape.  Development tool."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402
from tools.sweep_gemm import time_graph  # noqa: E402

PEAK_F16 = 2516.0   # dense fp16 MFMA TFLOP/s: 256 CU x 4096 flop/clk x 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="512x11008x4096,512x4096x4096,16x11008x4096,512x4096x16384")
    ap.add_argument("--cfgs", default="-1,0,1,2")
    args = ap.parse_args()
    lib = _capi.load()
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        g = torch.Generator().manual_seed(0)
        q = torch.randint(-128, 128, (K, N), generator=g, dtype=torch.int8).cuda()
        s = (torch.rand(N, generator=g) * 0.01 + 0.001).half().cuda()
        x = torch.randn(M, K, generator=g).half().cuda()
        wp = mixlib.PackW8A16(q)
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        ref = x.float() @ (q.float() * s.float())
        for c in [int(v) for v in args.cfgs.split(",")]:
            assert lib.mixq_gemm_w8a16_set_config(c) == 0
            y = mixlib.W8A16Linear(x, wp, s, None, N, K, out=out)
            err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
            us = time_graph(lambda: mixlib.W8A16Linear(x, wp, s, None, N, K, out=out), 50)
            tf = 2.0 * M * N * K / us / 1e6
            gbs = (N * K + 2 * M * K + 2 * M * N) / us / 1e3
            print(f"{shp} cfg{c:2d}: rel_err={err:.1e} {us:8.2f} us  {tf:7.1f} TFLOP/s ({100 * tf / PEAK_F16:4.1f}% of fp16 peak)  {gbs:7.1f} GB/s", flush=True)
        lib.mixq_gemm_w8a16_set_config(-1)


if __name__ == "__main__":
    main()
