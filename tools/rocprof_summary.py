#!/usr/bin/env python3
"""Summarise a rocprofv3 result database (rocpd sqlite): per-kernel call count / average duration, and per-kernel
average of every collected PMC counter (summed over instances per dispatch).  Output is what gets committed under
profiles/."""
import sqlite3
import sys
from collections import defaultdict


WIDTH = 70


def short(name, n=None):
    n = n or WIDTH
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    db = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    global WIDTH
    if len(sys.argv) > 3:
        WIDTH = int(sys.argv[3])                # full kernel names (library kernels encode their tiling in them)
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select name, start, end from kernels").fetchall()
    agg = defaultdict(list)
    for name, s, e in rows:
        agg[name].append(e - s)
    print(f"# {db}")
    # median_us: the kernel's duration in the steady state (the average also holds the first, clock-unsettled launches after every idle gap)
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'median_us':>10s} {'min_us':>10s} {'total_ms':>10s}")
    for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if filt and filt not in name:
            continue
        med = sorted(d)[len(d) // 2]
        print(f"{short(name):72s} {len(d):6d} {sum(d) / len(d) / 1e3:10.2f} {med / 1e3:10.2f} {min(d) / 1e3:10.2f} {sum(d) / 1e6:10.3f}")
    try:
        pm = cur.execute("select name, dispatch_id, counter_name, counter_value from pmc_events").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        per = defaultdict(lambda: defaultdict(float))
        for name, disp, cn, cv in pm:
            per[(name, disp)][cn] += cv
        ksum = defaultdict(lambda: defaultdict(list))
        for (name, disp), cs in per.items():
            for cn, v in cs.items():
                ksum[name][cn].append(v)
        print("\n# PMC counters: average per dispatch (sum over instances)")
        for name, cs in ksum.items():
            if filt and filt not in name:
                continue
            print(short(name, 110))
            for cn, vals in sorted(cs.items()):
                print(f"    {cn:34s} {sum(vals) / len(vals):18.1f}   (n={len(vals)})")


if __name__ == "__main__":
    main()
