#!/usr/bin/env python3
"""GPU-side development sweep: every GEMM tile config x a few shapes, checked against a torch fp64 matmul on the
same GPU (exact for int8 operands) and timed with events on the launch stream.  Not a parity test (those are in
tests/ against the CPU oracle); this is the tuning loop's measuring stick.  Writes gpurun_out/sweep_gemm.json.
"""
import argparse
import json
import os
import sys
import time

import torch
os.environ.setdefault("MIXQ_TUNING_LIB", "1")      # ablation kernels / trace stamps / probe knobs live in the tools build (make tuning)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402

PEAK_TOPS = 5033.0


def time_fn(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters  # us


def time_graph(fn, iters=50, unroll=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(unroll):
                fn()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters // unroll):
            g.replay()
        e.record()
        torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (iters // unroll * unroll)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="512x11008x4096")
    ap.add_argument("--cfgs", default="all")
    ap.add_argument("--bit", type=int, default=8)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--packed", default="0,1,2", help="operand formats to try: 0 plain, 1 P16x64, 2 weights F16x64 + activations P16x64")
    ap.add_argument("--out", default="gpurun_out/sweep_gemm.json")
    ap.add_argument("--krot", default="0", help="k-step rotation(s) between neighbouring N tiles for the wr* tilings")
    ap.add_argument("--nout", type=int, default=0, help="outlier columns fed to the fp16 tail (timing; the error check ignores them)")
    args = ap.parse_args()
    dev = "cuda"
    lib = _capi.load()
    print(_capi.device_info(), flush=True)
    _capi.ensure_workspace(dev)            # the stream-K configs (opt-in) need their hand-off workspace
    names = _capi.gemm_config_names()
    cfgs = list(range(len(names))) if args.cfgs == "all" else [int(c) for c in args.cfgs.split(",")]
    results = []
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        g = torch.Generator(device="cpu").manual_seed(0)
        if args.bit == 8:
            qx = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
            qw = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
            ref32 = (qx.double() @ qw.double().T)
        else:
            vx = torch.randint(-7, 8, (M, K), generator=g, dtype=torch.int8)
            vw = torch.randint(-8, 8, (N, K), generator=g, dtype=torch.int8)
            from mixq_amd.linear import pack_to_i4
            qx, qw = pack_to_i4(vx).to(dev), pack_to_i4(vw).to(dev)
            ref32 = (vx.to(dev).double() @ vw.to(dev).double().T)
        sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
        sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
        ref = (ref32 * sx.double() * sw.double())
        flops = 2.0 * M * N * K
        can_pack = (K if args.bit == 8 else K // 2) % 64 == 0
        packs = {0: (qx, qw)}
        if can_pack:
            packs[1] = (mixlib.PackOperand(qx, 1), mixlib.PackOperand(qw, 1))       # P16x64
            packs[2] = (packs[1][0], mixlib.PackOperand(qw, 2))                     # weights F16x64 (gemm_wreg.hip, decode32)
        xo = wo = None
        if args.nout:
            pad = (args.nout + 15) // 16 * 16
            xo = torch.zeros((M, pad), dtype=torch.float16, device=dev)[:, :args.nout]
            wo = torch.zeros((N, pad), dtype=torch.float16, device=dev)[:, :args.nout]
        for c, krot, rotv in [(c, int(kr), int(rv)) for c in cfgs for kr in args.packed.split(",") for rv in args.krot.split(",")]:
            if krot not in packs or (rotv and not names[c].startswith("wr")):
                continue
            if hasattr(lib, "mixq_gemm_set_krot"):
                lib.mixq_gemm_set_krot(rotv)
            if names[c].startswith("wr") != (krot == 2) and names[c] != "decode32":
                continue                                   # the weights-in-registers tilings take F16x64 only, the others never
            if names[c] == "decode32" and (krot == 0 or M > 32):
                continue
            ax, aw = packs[krot]
            pk = {}
            rc = lib.mixq_gemm_set_config(c)
            assert rc == 0
            try:
                y = mixlib.FusedLinear(ax, aw, sx, sw, xo, wo, args.nout, None, M, N, K, bit=args.bit, **pk)
                torch.cuda.synchronize()
            except Exception as ex:  # config incompatible with the shape
                print(f"{shp} cfg{c} {names[c]}: {ex}", flush=True)
                continue
            err = (y.double() - ref).abs().max().item()
            rel = err / ref.abs().max().item()
            out = torch.empty_like(y)
            us = time_fn(lambda: mixlib.FusedLinear(ax, aw, sx, sw, xo, wo, args.nout, None, M, N, K, bit=args.bit, out=out, **pk), args.iters)
            usg = time_graph(lambda: mixlib.FusedLinear(ax, aw, sx, sw, xo, wo, args.nout, None, M, N, K, bit=args.bit, out=out, **pk), args.iters)
            tops = flops / usg / 1e6
            r = dict(shape=shp, cfg=c, krot=krot, rot=rotv, name=names[c], bit=args.bit, max_abs_err=err, rel_err=rel, us_eager=us, us_graph=usg,
                     tops=tops, frac=tops / PEAK_TOPS)
            results.append(r)
            print(f"{shp} cfg{c:2d} packed={krot:1d} rot={rotv:2d} {names[c]:24s} err={err:.3e} rel={rel:.2e} eager={us:8.2f}us graph={usg:8.2f}us "
                  f"{tops:7.1f} TOPS ({100 * tops / PEAK_TOPS:4.1f}%)", flush=True)
        lib.mixq_gemm_set_config(-1)
        if hasattr(lib, "mixq_gemm_set_krot"):
            lib.mixq_gemm_set_krot(0)
        for fmt in (1, 2):
            auto = lib.mixq_gemm_pick_config_fmt(M, N, K, args.bit, fmt)
            print(f"{shp}: auto pick (fmt {fmt}) = cfg{auto} {names[auto]}", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
