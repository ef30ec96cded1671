#!/usr/bin/env python3
"""Round-3 yardsticks for "what does this part sustain on an int8 GEMM" (Next #1a).  MEASUREMENT ONLY: nothing here is
on the product path.

For every shape, in ONE process and interleaved round-robin (cdna_hip_programming.md 5.4 rules 13 / 24 / 25: same random operands for
every arm, one hipGraph of `--launches` launches per arm, per-arm median over the rounds):
  int_mm     torch._int_mm(A[M,K] int8, W[N,K].t())  -> int32 [M,N]   (the vendor library: hipBLASLt behind PyTorch)
  fp16_mm    torch.mm(A.half(), W.half().t())        -> fp16          (context: the fp16 library GEMM of the same shape)
  mixq       this library's fused int8 GEMM (dequant epilogue, fp16 out, no outlier columns), auto-picked tiling
  mixq_t     the same with 41 / 1 % outlier columns (the metric's operator)
Then, per arm, a ~2 s loop of graph replays with the board power and shader clock sampled beside it (hwmon sysfs, rocm-smi fallback),
and the bare-MFMA micro-benchmark (tools/ubench_mfma SECONDS) under the same sampler.  Output -> profiles/r03_ceiling.txt."""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixq_amd import _capi, mixlib  # noqa: E402


class Sampler:
    """Board power (W) and shader clock (MHz) every `period` s from hwmon sysfs; rocm-smi --json when sysfs has neither."""

    def __init__(self, period=0.1):
        self.period = period
        self.power_f = self.clk_f = None
        # the box's sysfs lists every GPU of the host; the one HIP gave us is found by its PCI address
        want = None
        try:
            pr = torch.cuda.get_device_properties(0)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        except Exception:
            pass
        self.pci = want
        cards = glob.glob("/sys/class/drm/card*/device")
        mine = [c for c in cards if want and os.path.basename(os.path.realpath(c)).startswith(want)] or cards
        for h in sorted(sum((glob.glob(c + "/hwmon/hwmon*") for c in mine), [])):
            for nm in ("power1_average", "power1_input"):
                if self.power_f is None and os.path.exists(os.path.join(h, nm)):
                    self.power_f = os.path.join(h, nm)
            if self.clk_f is None and os.path.exists(os.path.join(h, "freq1_input")):
                self.clk_f = os.path.join(h, "freq1_input")
        self.samples = []
        self._stop = threading.Event()
        self._t = None

    def source(self):
        return f"pci={self.pci} power={self.power_f or 'rocm-smi'} sclk={self.clk_f or 'rocm-smi'}"

    def _read(self):
        p = c = None
        try:
            if self.power_f:
                p = int(open(self.power_f).read()) / 1e6
            if self.clk_f:
                c = int(open(self.clk_f).read()) / 1e6
        except (OSError, ValueError):
            pass
        if p is None or c is None:
            try:
                j = json.loads(subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=5).stdout)
                card = next(iter(j.values()))
                for k, v in card.items():
                    kl = k.lower()
                    if p is None and "power" in kl and "(w)" in kl:
                        p = float(v)
                    if c is None and "sclk" in kl and "level" not in kl:
                        c = float(str(v).strip("()Mhz "))
            except Exception:
                pass
        return p, c

    def start(self):
        self.samples = []
        self._stop.clear()

        def loop():
            while not self._stop.is_set():
                self.samples.append((time.time(),) + self._read())
                time.sleep(self.period)
        self._t = threading.Thread(target=loop, daemon=True)
        self._t.start()

    def stop(self, skip=0.7):
        self._stop.set()
        self._t.join()
        if not self.samples:
            return "no samples"
        t0 = self.samples[0][0]
        late = [s for s in self.samples if s[0] - t0 >= skip] or self.samples
        ps = [s[1] for s in late if s[1] is not None]
        cs = [s[2] for s in late if s[2] is not None]
        f = lambda v: f"{np.mean(v):7.1f} (max {np.max(v):7.1f})" if v else "   n/a"
        return f"power W {f(ps)}   sclk MHz {f(cs)}   [{len(late)} samples]"


def energy_table(args, smp, side):
    """where do the joules go?  Per arm ~`--power-seconds` of back-to-back graph replays under the power sampler:
    us per launch, mean board power, energy per launch (mJ) and TOPS/W.  The ablation arms compute garbage by design (no weight loads /
    no activation traffic / neither) - they price the feeds, not the result."""
    dev = "cuda"
    lib = _capi.load()
    names = _capi.gemm_config_names()
    forced = [nm for nm in ("wr128x192_s16_d4_l2", "wr128x192_abl1_noW", "wr128x192_abl2_noX", "wr128x192_abl3_mfma") if nm in names]
    if len(forced) < 4:
        print("(the ablation kernels are only in the tuning build: run with MIXQ_TUNING_LIB=1)")
    idle = None
    smp.start(); time.sleep(1.0)
    idle_s = smp.stop(skip=0.0)
    print("idle:", idle_s)
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        g = torch.Generator().manual_seed(0)
        xf = torch.randn(M, K, generator=g)
        qx = torch.round(xf / (xf.abs().amax(dim=1, keepdim=True) / 127)).to(torch.int8).to(dev)
        qw = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
        sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
        sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
        xp, wp = mixlib.PackOperand(qx, 1), mixlib.PackOperand(qw, 2)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        wt = qw.t()
        arms = [("hipBLASLt int8 (torch._int_mm, int32 out)", -1, lambda: torch._int_mm(qx, wt)),
                ("this library, automatic tiling", -1, lambda: mixlib.FusedLinear(xp, wp, sx, sw, None, None, 0, None, M, N, K, bit=8, out=out))]
        for nm in forced:
            arms.append((nm, names.index(nm), lambda: mixlib.FusedLinear(xp, wp, sx, sw, None, None, 0, None, M, N, K, bit=8, out=out)))
        flops = 2.0 * M * N * K
        print(f"{shp}: ~{args.power_seconds:.0f} s of back-to-back replays per arm ({args.launches} launches per graph); board power from {smp.source()}")
        print(f"  {'arm':44s} {'us/launch':>10s} {'TOPS':>8s} {'W':>8s} {'sclk MHz':>9s} {'mJ/launch':>10s} {'TOPS/W':>7s}")
        with torch.cuda.stream(side):
            for label, cfg, fn in arms:
                lib.mixq_gemm_set_config(cfg)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    for _ in range(args.launches):
                        fn()
                torch.cuda.synchronize()
                lib.mixq_gemm_set_config(-1)
                for _ in range(200):                                   # clock conditioning in front of the sampled window
                    gr.replay()
                torch.cuda.synchronize()
                smp.start()
                t0 = time.time(); n = 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                while time.time() - t0 < args.power_seconds:
                    for _ in range(50):
                        gr.replay()
                    n += 50
                    torch.cuda.synchronize()
                e1.record(side); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (n * args.launches)
                smp._stop.set(); smp._t.join()
                t00 = smp.samples[0][0]
                late = [x for x in smp.samples if x[0] - t00 >= 0.5] or smp.samples
                pw = float(np.mean([x[1] for x in late if x[1] is not None])) if any(x[1] is not None for x in late) else float("nan")
                ck = float(np.mean([x[2] for x in late if x[2] is not None])) if any(x[2] is not None for x in late) else float("nan")
                tops = flops / us / 1e6
                print(f"  {label:44s} {us:10.2f} {tops:8.1f} {pw:8.1f} {ck:9.0f} {pw * us * 1e-3:10.2f} {tops / pw:7.2f}", flush=True)
                del gr
    lib.mixq_gemm_set_config(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="512x11008x4096,2048x11008x4096,4096x11008x4096,512x28672x8192,512x4096x4096")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--power-seconds", type=float, default=2.0)
    ap.add_argument("--no-power", action="store_true")
    ap.add_argument("--energy", action="store_true", help="energy-per-launch table (board power x time) for the vendor int8 GEMM, this library's automatic "
                    "choice, the 128x192 weights-in-registers tile and its feed ablations (needs MIXQ_TUNING_LIB=1); replaces the default report")
    args = ap.parse_args()
    dev = "cuda"
    print("device:", _capi.device_info())
    print("torch", torch.__version__, "hip", torch.version.hip)
    smp = Sampler()
    print("sampler:", smp.source(), "idle:", end=" ")
    smp.start(); time.sleep(0.5); print(smp.stop(skip=0.0))
    side = torch.cuda.Stream()
    if args.energy:
        return energy_table(args, smp, side)
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        g = torch.Generator().manual_seed(0)
        # the bench's operand statistics: activations quantised from N(0,1) rows (|q| <= 127, typical 30), weights uniform
        xf = torch.randn(M, K, generator=g)
        qx = torch.round(xf / (xf.abs().amax(dim=1, keepdim=True) / 127)).to(torch.int8).to(dev)
        qw = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
        sx = (torch.rand(M, 1, generator=g) * 0.01 + 0.001).half().to(dev)
        sw = (torch.rand(1, N, generator=g) * 0.01 + 0.001).half().to(dev)
        xp, wp = mixlib.PackOperand(qx, 1), mixlib.PackOperand(qw, 2)
        n_out = max(1, round(0.01 * K))
        pad = (n_out + 15) // 16 * 16
        xo = torch.randn((M, pad), device=dev).half()[:, :n_out]
        wo = torch.randn((N, pad), device=dev).half()[:, :n_out]
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        xh, wh = qx.half(), qw.half()
        wt = qw.t()
        arms = {
            "int_mm": lambda: torch._int_mm(qx, wt),
            "fp16_mm": lambda: torch.mm(xh, wh.t()),
            "mixq": lambda: mixlib.FusedLinear(xp, wp, sx, sw, None, None, 0, None, M, N, K, bit=8, out=out),
            "mixq_t": lambda: mixlib.FusedLinear(xp, wp, sx, sw, xo, wo, n_out, None, M, N, K, bit=8, out=out),
        }
        graphs = {}
        with torch.cuda.stream(side):
            for nm, fn in list(arms.items()):
                try:
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=side):
                        for _ in range(args.launches):
                            fn()
                    torch.cuda.synchronize()
                    graphs[nm] = gr
                except Exception as e:                                  # a yardstick the library does not offer is reported, not fatal
                    print(f"  {shp} {nm}: unavailable ({type(e).__name__}: {str(e)[:120]})")
            # correctness of the yardstick against ours (int32 product through the fused kernel with unit scales is not comparable in
            # fp16 range; compare int_mm with a torch reference on a corner instead)
            if "int_mm" in graphs:
                ref = (qx[:64].float() @ qw[:64].float().t()).to(torch.int32)
                got = torch._int_mm(qx, wt)[:64, :64]
                assert torch.equal(ref, got), "torch._int_mm disagrees with a float reference"
            times = {nm: [] for nm in graphs}
            for r in range(args.rounds + 2):
                for nm, gr in graphs.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side); gr.replay(); e1.record(side)
                    torch.cuda.synchronize()
                    if r >= 2:
                        times[nm].append(e0.elapsed_time(e1) * 1e3 / args.launches)
            flops = 2.0 * M * N * K
            print(f"{shp}: {args.rounds} interleaved rounds x {args.launches} launches; us per launch median / min; TOPS at the median (% of 5033; fp16: % of 2516)")
            for nm in graphs:
                t = np.array(times[nm])
                peak = 2516.0 if nm == "fp16_mm" else 5033.0
                print(f"  {nm:8s} {np.median(t):8.2f} {t.min():8.2f}   {flops / np.median(t) / 1e6:7.1f} ({100 * flops / np.median(t) / 1e6 / peak:4.1f} %)", flush=True)
            if not args.no_power:
                for nm, gr in graphs.items():
                    smp.start()
                    t0 = time.time(); n = 0
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                    while time.time() - t0 < args.power_seconds:
                        for _ in range(20):
                            gr.replay()
                        n += 20
                        torch.cuda.synchronize()
                    e1.record(side); torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / (n * args.launches)
                    print(f"  {nm:8s} sustained {us:8.2f} us = {flops / us / 1e6:7.1f} TOPS   {smp.stop()}", flush=True)
        del graphs
    ub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_mfma")
    if os.path.exists(ub) and not args.no_power:
        print("bare MFMA (tools/ubench_mfma: constant register operands, no memory traffic - an upper bound on the clock):")
        smp.start()
        r = subprocess.run([ub, "3"], capture_output=True, text=True)
        print("  " + r.stdout.strip().replace("\n", "\n  "))
        print("  over the whole run:", smp.stop())


if __name__ == "__main__":
    main()
