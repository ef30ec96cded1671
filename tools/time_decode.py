#!/usr/bin/env python3
"""Small-batch (decode) timing of the fused W8A8 GEMM with COLD weights: the launches of one graph rotate through enough
weight copies to exceed the 256 MB MALL, as a model's layers do.  Prints us per launch and weight-stream TB/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="1x11008x4096,16x11008x4096,32x11008x4096,16x4096x4096,16x4096x11008")
    ap.add_argument("--cfgs", default="-1,12")
    ap.add_argument("--copies", type=int, default=10)
    ap.add_argument("--bit", type=int, default=8)
    args = ap.parse_args()
    lib = _capi.load()
    names = _capi.gemm_config_names()
    dev = "cuda"
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        g = torch.Generator().manual_seed(0)
        KB = K if args.bit == 8 else K // 2
        qx = torch.randint(-127, 128, (M, KB), generator=g, dtype=torch.int8).to(dev)
        sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
        sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
        ws = [mixlib.PackP16x64(torch.randint(-127, 128, (N, KB), generator=g, dtype=torch.int8).to(dev)) for _ in range(args.copies)]
        qxp = mixlib.PackP16x64(qx)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        for c in [int(v) for v in args.cfgs.split(",")]:
            assert lib.mixq_gemm_set_config(c) == 0
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for w in ws:
                    mixlib.FusedLinear(qxp, w, sx, sw, None, None, 0, None, M, N, K, bit=args.bit, out=out, x_packed=True, w_packed=True)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(3):
                        for w in ws:
                            mixlib.FusedLinear(qxp, w, sx, sw, None, None, 0, None, M, N, K, bit=args.bit, out=out, x_packed=True, w_packed=True)
                torch.cuda.synchronize()
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    gr.replay()
                e1.record()
                torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (5 * 3 * len(ws))
            name = "auto" if c < 0 else names[c]
            print(f"{shp} {name:24s} {us:8.2f} us  {N * KB / us / 1e6:6.2f} TB/s of weights (cold)", flush=True)
        lib.mixq_gemm_set_config(-1)


if __name__ == "__main__":
    main()
