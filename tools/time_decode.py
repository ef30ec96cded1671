#!/usr/bin/env python3
"""Small-batch (decode) timing of the fused W8A8 / W4A4 GEMM, interleaved A/B over configurations.  COLD: the launches of one graph
rotate through enough weight copies to exceed the 256 MB MALL, as a model's layers do; WARM (--copies 1): one layer replayed.
Prints us per launch and the weight-stream rate."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="32x11008x4096,16x11008x4096,32x4096x4096,32x4096x11008")
    ap.add_argument("--cfgs", default="decode32,wr64x64_s8_d4_l1")
    ap.add_argument("--copies", type=int, default=12)
    ap.add_argument("--bit", type=int, default=8)
    ap.add_argument("--nout", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=10)
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
        xo = wo = None
        if args.nout:
            pad = (args.nout + 15) // 16 * 16
            xo = torch.randn((M, pad), device=dev).half()[:, :args.nout]
            wo = torch.randn((N, pad), device=dev).half()[:, :args.nout]
        raw = [torch.randint(-127, 128, (N, KB), generator=g, dtype=torch.int8).to(dev) for _ in range(args.copies)]
        ws = {f: [mixlib.PackOperand(w, f) for w in raw] for f in (1, 2)}
        del raw
        qxp = mixlib.PackOperand(qx, 1)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        graphs = []
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for nm in args.cfgs.split(","):
                c = -1 if nm == "auto" else names.index(nm)
                fmt = 2 if nm.startswith("wr") or nm == "auto" else 1
                assert lib.mixq_gemm_set_config(c) == 0
                run = lambda w: mixlib.FusedLinear(qxp, w, sx, sw, xo, wo, args.nout, None, M, N, K, bit=args.bit, out=out)
                for w in ws[fmt]:
                    run(w)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(2):
                        for w in ws[fmt]:
                            run(w)
                torch.cuda.synchronize()
                graphs.append((nm, gr))
            lib.mixq_gemm_set_config(-1)
            times = {nm: [] for nm, _ in graphs}
            for r in range(args.rounds + 2):
                for nm, gr in graphs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gr.replay()
                    e1.record()
                    e1.synchronize()
                    if r >= 2:
                        times[nm].append(e0.elapsed_time(e1) * 1e3 / (2 * args.copies))
        print(f"{shp} bit={args.bit} n_out={args.nout} copies={args.copies} ({'cold' if args.copies * N * KB > 256e6 else 'warm'}): us per launch median / min", flush=True)
        for nm, _ in graphs:
            t = np.array(times[nm])
            med = float(np.median(t))
            print(f"  {nm:28s} {med:8.2f} {t.min():8.2f}   {N * KB / med / 1e6:6.2f} TB/s of weights", flush=True)


if __name__ == "__main__":
    main()
