#!/usr/bin/env python3
"""Per-workgroup timeline of the fused GEMM from the kernel's own timestamps (mixq_gemm_set_trace, 100 MHz clock).
Prints, over the workgroups of one launch, the distribution of each phase: launch skew (entry relative to the first
workgroup's entry), first stage landed, k loop, epilogue arithmetic, store issue, store retire.  Development tool."""
import argparse
import os
import sys

import numpy as np
import torch
os.environ.setdefault("MIXQ_TUNING_LIB", "1")      # ablation kernels / trace stamps / probe knobs live in the tools build (make tuning)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="512x11008x4096")
    ap.add_argument("--cfgs", default="8", help="configuration indices or names, comma-separated")
    ap.add_argument("--bit", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--nout", type=int, default=0, help="outlier columns (the fp16 MFMA tail of the epilogue)")
    ap.add_argument("--cold", type=int, default=0, help="rotate over this many copies of the weight image (8 x 45 MB > 256 MB memory-side cache): the traced launch streams its weights from HBM")
    ap.add_argument("--panels", action="store_true", help="also print the panel pipeline of the epilogue (consumer wave 0 and loader wave 0 stamps)")
    ap.add_argument("--f6", action="store_true", help="bit 4 with both operands as FP6 codes (MIXQ_FMT_F6X128): the FP6-pipe form of the wr kernels")
    args = ap.parse_args()
    dev = "cuda"
    lib = _capi.load()
    names = _capi.gemm_config_names()
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        g = torch.Generator(device="cpu").manual_seed(0)
        KB = K if args.bit == 8 else K // 2
        qx = torch.randint(-127, 128, (M, KB), generator=g, dtype=torch.int8).to(dev)
        qw = torch.randint(-127, 128, (N, KB), generator=g, dtype=torch.int8).to(dev)
        sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
        sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
        if args.bit == 4:
            qx, qw = qx.view(torch.uint8), qw.view(torch.uint8)
        qxp = mixlib.PackOperand(qx, 4 if args.f6 else 1)
        qw_by_fmt = {1: mixlib.PackOperand(qw, 1), 2: mixlib.PackOperand(qw, 3 if args.f6 else 2)}
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        xo = wo = None
        if args.nout:
            pad = (args.nout + 15) // 16 * 16
            xo = torch.randn((M, pad), device=dev).half()[:, :args.nout]
            wo = torch.randn((N, pad), device=dev).half()[:, :args.nout]
        trace = torch.zeros(2 * 16 * 4096, dtype=torch.int64, device=dev)
        for c in [int(v) if v.lstrip("-").isdigit() else names.index(v) for v in args.cfgs.split(",")]:
            assert lib.mixq_gemm_set_config(c) == 0
            qwp = qw_by_fmt[2 if names[c].startswith("wr") else 1]
            copies = [qwp] + [mixlib.set_fmt(qwp.clone(), mixlib.fmt_of(qwp)) for _ in range(max(1, args.cold) - 1)]
            state = {"i": 0}
            def run():
                state["i"] += 1
                return mixlib.FusedLinear(qxp, copies[state["i"] % len(copies)], sx, sw, xo, wo, args.nout, None, M, N, K, bit=args.bit, out=out)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            lib.mixq_gemm_set_trace(trace.data_ptr())
            rows = []
            for _ in range(args.reps):
                trace.zero_()
                for _ in range(2 if not args.cold else 2 * len(copies)):
                    run()               # back to back: the last launch overwrites the earlier ones (steady-state clocks)
                torch.cuda.synchronize()
                full = trace.cpu().numpy()
                t = full[:16 * 4096].reshape(-1, 16)
                live = t[:, 0] != 0
                t2 = full[16 * 4096:].reshape(-1, 16)[live].astype(np.float64)
                t = t[live].astype(np.float64)
                rows.append(t)
            lib.mixq_gemm_set_trace(None)
            t = rows[-1]
            t0 = t[:, 0].min()
            ph = {
                "entry skew": t[:, 0] - t0,
                "entry->stage0": t[:, 1] - t[:, 0],
                "k loop": t[:, 2] - t[:, 1],
                "  loop end->barrier": t[:, 6] - t[:, 2],
                "  dequant+stage": t[:, 7] - t[:, 6],
                "  ->barrier 2": t[:, 3] - t[:, 7],
                "epilogue math": t[:, 3] - t[:, 2],
                "store issue": t[:, 4] - t[:, 3],
                "store retire": t[:, 5] - t[:, 4],
                "WG total": t[:, 5] - t[:, 0],
                "end (vs first entry)": t[:, 5] - t0,
            }
            mhz = (t[:, 10] - t[:, 9]) / ((t[:, 2] - t[:, 1]) / 100.0)
            cyc = np.median(t[:, 10] - t[:, 9]) / (K if args.bit == 8 else K // 2) * 64
            print(f"{shp} bit={args.bit} n_out={args.nout}{' COLD weights' if args.cold else ''} cfg{c} {names[c]}: {t.shape[0]} workgroups; s_memtime ticks per us in the k loop: "
                  f"{np.median(mhz):.0f} ({cyc:.0f} shader cycles per 64-byte k-step); microseconds min / median / p90 / max")
            for k, v in ph.items():
                v = v / 100.0
                print(f"  {k:22s} {v.min():7.2f} {np.median(v):7.2f} {np.percentile(v, 90):7.2f} {v.max():7.2f}")
            if args.panels and t2[:, 0].any():
                b1 = t[:, 6]                                   # consumer wave 0 past the epilogue's first barrier
                print("  panel pipeline, microseconds after the first epilogue barrier (median): consumer wave 0 has staged the panel; loader wave 0 saw its flags / issued its copy")
                for pn in range(4):
                    line = f"    panel {pn}: staged {np.median(t2[:, 2 * pn] - b1) / 100:5.2f}"
                    if t2[:, 9 + 2 * pn].any() and pn < 3:
                        l0, l1 = np.median(t2[:, 9 + 2 * pn] - b1) / 100, np.median(t2[:, 10 + 2 * pn] - b1) / 100
                        line += f"   loader: flags seen {l0:5.2f}  copy issued {l1:5.2f}"
                    print(line)
        lib.mixq_gemm_set_config(-1)


if __name__ == "__main__":
    main()
