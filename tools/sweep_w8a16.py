#!/usr/bin/env python3
"""Time the weight-only W8A16 GEMM per config and shape: one hipGraph of `--launches` launches per configuration, replayed
round-robin for `--rounds` rounds (interleaved A/B), median / min per configuration.  Development tool."""
import argparse
import os
import sys

import numpy as np
import torch
os.environ.setdefault("MIXQ_TUNING_LIB", "1")      # ablation kernels / trace stamps / probe knobs live in the tools build (make tuning)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402

PEAK_F16 = 2516.0   # dense fp16 MFMA TFLOP/s: 256 CU x 4096 flop/clk x 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="512x11008x4096,512x4096x4096,512x4096x11008,16x11008x4096")
    ap.add_argument("--cfgs", default="auto", help="comma list of config names, 'auto', or 'all'")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--launches", type=int, default=20)
    args = ap.parse_args()
    lib = _capi.load()
    names = _capi.w8a16_config_names()
    side = torch.cuda.Stream()
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        g = torch.Generator().manual_seed(0)
        q = torch.randint(-128, 128, (K, N), generator=g, dtype=torch.int8).cuda()
        s = (torch.rand(N, generator=g) * 0.01 + 0.001).half().cuda()
        x = torch.randn(M, K, generator=g).half().cuda()
        wp = mixlib.PackW8A16(q)
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        ref = x.float() @ (q.float() * s.float())
        if args.cfgs == "all":
            todo = ["auto"] + [nm for nm in names if (M <= 32) == nm.endswith("decode32") or "abl" not in nm and M <= 32]
        else:
            todo = args.cfgs.split(",")
        graphs, errs = [], {}
        with torch.cuda.stream(side):
            for nm in todo:
                c = -1 if nm == "auto" else names.index(nm)
                if nm.endswith("decode32") and M > 32:
                    continue
                assert lib.mixq_gemm_w8a16_set_config(c) == 0
                run = lambda: mixlib.W8A16Linear(x, wp, s, None, N, K, out=out)
                for _ in range(3):
                    y = run()
                torch.cuda.synchronize()
                errs[nm] = (y.float() - ref).abs().max().item() / ref.abs().max().item()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    for _ in range(args.launches):
                        run()
                torch.cuda.synchronize()
                graphs.append((nm, gr))
            lib.mixq_gemm_w8a16_set_config(-1)
            times = {nm: [] for nm, _ in graphs}
            for r in range(args.rounds + 2):
                for nm, gr in graphs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gr.replay()
                    e1.record()
                    e1.synchronize()
                    if r >= 2:
                        times[nm].append(e0.elapsed_time(e1) * 1e3 / args.launches)
        print(f"{shp}: {args.rounds} interleaved rounds x {args.launches} launches; us per launch median / min", flush=True)
        for nm, _ in graphs:
            t = np.array(times[nm])
            med = float(np.median(t))
            tf = 2.0 * M * N * K / med / 1e6
            print(f"  {nm:28s} {med:8.2f} {t.min():8.2f}   {tf:7.1f} TFLOP/s ({100 * tf / PEAK_F16:4.1f} % of fp16 peak)  rel_err {errs[nm]:.1e}", flush=True)


if __name__ == "__main__":
    main()
