#!/usr/bin/env python3
"""Where a step's time goes under the DRIVER's invocation of bench.py (`--gpus 1 --steps 20 --warmup 5`), from a rocprofv3
--kernel-trace database of that very command: kernels are grouped into bursts (one burst = one graph replay: consecutive kernels less
than GAP_US apart), and for every burst of exactly 2 x steps kernels (quantise + GEMM per step) the span is split into kernel time
(quantise, GEMM) and inter-kernel gaps (quantise -> GEMM inside a step, GEMM -> next step's quantise).  VERDICT r04 item 1(a).

usage: driver_gaps.py <rocpd .db> [steps]"""
import sqlite3
import sys

GAP_US = 25.0


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


def main():
    db = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    ks = [(n.replace("(anonymous namespace)::", ""), s / 1e3, e / 1e3) for n, s, e in rows]
    # runs of strictly alternating quantise / GEMM kernels less than GAP_US apart (a replay's restore copy, the GEMM-only graph and the
    # eager self-check end a run); a run of r x steps pairs is r back-to-back replays and is cut into them
    runs, cur = [], []
    for k in ks:
        want_q = len(cur) % 2 == 0
        ok = ("quant" in k[0]) if want_q else ("gemm" in k[0])
        if cur and (not ok or k[1] - cur[-1][2] > GAP_US):
            runs.append(cur[: len(cur) // 2 * 2])
            cur = []
            ok = "quant" in k[0]
        if ok:
            cur.append(k)
    if cur:
        runs.append(cur[: len(cur) // 2 * 2])
    bursts = []
    for r in runs:
        if len(r) >= 2 * steps and len(r) % (2 * steps) == 0:
            bursts += [r[i:i + 2 * steps] for i in range(0, len(r), 2 * steps)]
    print(f"# {db}: {len(ks)} kernels; {len(bursts)} graph replays of {steps} steps found (runs of alternating quantise / GEMM kernels < {GAP_US:.0f} us apart)")
    full = []
    for i, b in enumerate(bursts):
        q = sum(b[j][2] - b[j][1] for j in range(0, len(b), 2))
        g = sum(b[j][2] - b[j][1] for j in range(1, len(b), 2))
        gap_qg = sum(b[j + 1][1] - b[j][2] for j in range(0, len(b), 2))
        gap_gq = sum(b[j + 1][1] - b[j][2] for j in range(1, len(b) - 1, 2))
        span = b[-1][2] - b[0][1]
        before = b[0][1] - bursts[i - 1][-1][2] if i else float("nan")
        full.append((i, span, q, g, gap_qg, gap_gq, before))
    if not full:
        print("no run of", 2 * steps, "alternating quantise / GEMM kernels found")
        return
    print(f"# {len(full)} graph replays of {steps} steps (quantise + GEMM each); per STEP, us")
    print(f"{'':64s} {'span':>8s} {'quant':>8s} {'gemm':>8s} {'gap q>g':>8s} {'gap g>q':>8s} {'idle before the replay (us)':>30s}")

    def line(tag, sel):
        if not sel:
            return
        print(f"{tag:64s} {med([f[1] for f in sel]) / steps:8.3f} {med([f[2] for f in sel]) / steps:8.3f} {med([f[3] for f in sel]) / steps:8.3f} "
              f"{med([f[4] for f in sel]) / steps:8.3f} {med([f[5] for f in sel]) / (steps - 1):8.3f} {med([f[6] for f in sel]):30.1f}")

    line("first replay", full[:1])
    line("all replays (median)", full)
    n = len(full)
    line("last third (median)", full[2 * n // 3:])
    spans = sorted(f[1] / steps for f in full)
    print(f"span per step over the replays: min {spans[0]:.3f}  p25 {spans[n // 4]:.3f}  median {spans[n // 2]:.3f}  p75 {spans[3 * n // 4]:.3f}  max {spans[-1]:.3f}")
    back = [f for f in full if f[6] < 200.0]
    idle = [f for f in full if not f[6] < 200.0]
    line("replays queued back to back (< 200 us behind the previous one)", back)
    line("replays from an idle stream", idle)
    # per-position profile inside a replay: is the first step of a replay slower than the rest?
    pos_q = [[] for _ in range(steps)]
    pos_g = [[] for _ in range(steps)]
    for i, *_ in full:
        b = bursts[i]
        for s in range(steps):
            pos_q[s].append(b[2 * s][2] - b[2 * s][1])
            pos_g[s].append(b[2 * s + 1][2] - b[2 * s + 1][1])
    print("kernel duration by position in the replay (median us): step: quant / gemm")
    print("  " + "  ".join(f"{s}: {med(pos_q[s]):.2f}/{med(pos_g[s]):.2f}" for s in range(steps)))


if __name__ == "__main__":
    main()
