#!/usr/bin/env python3
"""In-k-step timeline of the W4A4 (FP6-pipe) GEMM's main loop: s_memtime stamps taken inside ONE k-step (k-step nk / 2) by consumer wave 0
and loader wave 0 of every workgroup (tuning config wr128x192_f6r_t36_kstep_timeline; gemm_wreg.hip, ABLK 36).  Prints, in shader cycles
relative to the consumer's release from the k-step's barrier, the median / p10 / p90 over the workgroups of:
  consumer: reaches the weight wait | weights landed | barrier released | MFMA group j issued and its fragment re-read requested (j = 0..7)
  loader:   starts issuing the stage's DMA pieces | pieces issued | stage kt+1 landed (its vmcnt wait) | barrier released
Development tool: which wait do the consumers sit at."""
import argparse
import os
import sys

import numpy as np
import torch
os.environ.setdefault("MIXQ_TUNING_LIB", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402
from mixq_amd.linear import pack_to_i4  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="512x11008x4096")
    ap.add_argument("--cfg", default="wr128x192_f6r_t36_kstep_timeline", help="comma list of stamped configurations")
    ap.add_argument("--bit", type=int, default=4, help="4: W4A4 on the FP6 pipe; 8: the int8 loop (stamped configs wr128x192_t36_kstep_timeline / _t_wave2)")
    ap.add_argument("--ref", default="wr128x192_s16_d4_l2", help="the un-stamped configuration, timed beside it (what the stamps cost)")
    ap.add_argument("--nout", type=int, default=128)
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    dev = "cuda"
    lib = _capi.load()
    names = _capi.gemm_config_names()
    M, N, K = (int(v) for v in args.shape.split("x"))
    g = torch.Generator(device="cpu").manual_seed(0)
    if args.bit == 4:
        wf = mixlib.PackOperand(pack_to_i4(torch.randint(-8, 8, (N, K), generator=g, dtype=torch.int8)).to(dev), 3)
        xp = mixlib.PackOperand(pack_to_i4(torch.randint(-7, 8, (M, K), generator=g, dtype=torch.int8)).to(dev), 4)
    else:
        wf = mixlib.PackOperand(torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev), 2)
        xp = mixlib.PackOperand(torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(dev), 1)
    sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
    sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
    pad = (max(args.nout, 1) + 15) // 16 * 16
    xo = torch.randn((M, pad), device=dev).half()[:, :args.nout] if args.nout else None
    wo = torch.randn((N, pad), device=dev).half()[:, :args.nout] if args.nout else None
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    run = lambda: mixlib.FusedLinear(xp, wf, sx, sw, xo, wo, args.nout, None, M, N, K, bit=args.bit, out=out)

    def timed(cfg, n=20, rounds=9):
        assert lib.mixq_gemm_set_config(names.index(cfg)) == 0
        for _ in range(3):
            run()
        ts = []
        for _ in range(rounds):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                run()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3 / n)
        return float(np.median(ts))

    for cfg in args.cfg.split(","):
        t_ref, t_cfg = timed(args.ref), timed(cfg)
        ref_out = None
        assert lib.mixq_gemm_set_config(names.index(args.ref)) == 0
        run(); torch.cuda.synchronize(); ref_out = out.clone()
        assert lib.mixq_gemm_set_config(names.index(cfg)) == 0
        run(); torch.cuda.synchronize()
        print(f"{args.shape} {'W4A4 (FP6 pipe)' if args.bit == 4 else 'W8A8 (int8)'} n_out={args.nout}: {args.ref} {t_ref:.2f} us per launch (eager, back to back), {cfg} {t_cfg:.2f} us; outputs bit-identical: {torch.equal(ref_out, out)}")
        trace = torch.zeros(3 * 16 * 4096, dtype=torch.int64, device=dev)
        lib.mixq_gemm_set_trace(trace.data_ptr())
        rows = []
        for _ in range(args.reps):
            trace.zero_()
            run(); run(); run()
            torch.cuda.synchronize()
            t = trace.cpu().numpy()[2 * 16 * 4096:].reshape(-1, 16)
            t = t[t[:, 2] != 0].astype(np.float64)
            rows.append(t)
        lib.mixq_gemm_set_trace(None)
        lib.mixq_gemm_set_config(-1)
        t = np.concatenate(rows[2:])                          # (the first repetitions: clocks still ramping)
        ok = np.all(np.diff(t[:, :11], axis=1) >= 0, axis=1) & (t[:, 11:15].min(axis=1) > 0)
        print(f"  {t.shape[0]} workgroup samples, {int(ok.sum())} with monotonic stamps (others dropped)")
        t = t[ok]
        rel = t - t[:, 2:3]
        labels = ["C reaches weight wait", "C weights landed", "C barrier released"] + [f"C group {j} issued (+ re-read requested)" for j in range(8)] + \
                 ["L starts the stage's DMA", "L pieces issued", "L stage kt+1 landed", "L barrier released"]
        print("  shader cycles relative to the consumer's release from the k-step's barrier: p10 / median / p90")
        for i, lab in enumerate(labels):
            v = rel[:, i]
            print(f"    {lab:42s} {np.percentile(v, 10):8.0f} {np.median(v):8.0f} {np.percentile(v, 90):8.0f}")
        d = np.diff(t[:, 2:11], axis=1)
        print("  cycles per MFMA group (3 MFMAs; floor 3 x 19.5 = 58.5 FP6 / 3 x 16 = 48 int8), median: " + " ".join(f"{np.median(d[:, j]):.0f}" for j in range(8)))
        print(f"  k-step as the consumer sees it (weight wait of this k-step -> last group issued): median {np.median(t[:, 10] - t[:, 0]):.0f} cycles;"
              f" of which parked at the weight wait {np.median(t[:, 1] - t[:, 0]):.0f}, at the barrier {np.median(t[:, 2] - t[:, 1]):.0f}")
        print(f"  loader: issue of the stage's pieces {np.median(t[:, 12] - t[:, 11]):.0f} cycles, wait for stage kt+1 {np.median(t[:, 13] - t[:, 12]):.0f}, at the barrier {np.median(t[:, 14] - t[:, 13]):.0f}")


if __name__ == "__main__":
    main()
