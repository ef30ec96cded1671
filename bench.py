#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

metric  : effective int8 TFLOPS (= 2*M*N*K / t, outlier FLOPs and quantise work not counted, the reference's
          convention examples/benchbitsand.py:554-555) of the W8A8O16 MixQ Linear, (batch x seq, hidden) = (512, 4096)
          -> 11008, with 1 % synthetic outlier columns and outlier prediction frozen after 2 warm-up forwards.
step    : ONE full forward of MixLinear_GEMM over one 512-token batch already resident in HBM:
          (i)+(ii) fused extract/zero/absmax/quantise kernel  +  (iii)+(iv) int8 MFMA GEMM with fused dequant,
          fp16 outlier tail and (absent for Llama) bias.  Each step gets its own pristine activation buffer (the
          operator zeroes outlier columns in place, as the reference does), so no restore copy sits in the timed region.
timing  : the K steps are ONE hipGraph; barrier + synchronize on both sides of the timed region; the clock is HIP events on the launch
          stream (the host wall time around the same region is reported beside it); MAX over ranks.
          The metric is SUSTAINED throughput, and an MI355X needs 15-25 ms of continuous work to reach the clock it sustains
          (measured: the same graph replayed from idle runs 33-34 us per step for its first 2-3 ms, 30.9 us from ~12 ms on), so
          after the W warm-up steps and before the timed region the captured graph is replayed, untimed, for ~40 ms
          (CLOCK_SETTLE_MS; the inputs are restored afterwards).  One replay of a 20-step graph is a 0.6 ms sample that also carries
          the graph-launch latency (~45 us per replay from an idle stream: BENCH_r04's eager loop beat its own graph figure), so the
          timed region is --replays (31) back-to-back replays of the K-step graph, each between its own pair of events with the device
          never idle between them, and ms_per_step = MEDIAN replay / K (timing.replay_ms_min / median / max carry the spread; the
          single-replay figure of rounds 1-4 and the first replay after idle are reported beside it).  A step computes what it always did.
N GPUs  : one process per GPU, no data-path collective, one all_gather of {elapsed, flops} at the end.
          --scaling weak (default): every rank runs the same K steps on its own 512-token batches;
          --scaling strong: the 512 rows are split over the ranks (bench.shard_rows), weights replicated.
          value = sum of FLOPs over ranks / max time.  `python bench.py --gpus N` launches its N ranks itself; under
          `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it uses the ranks it was given.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--shape K,N] [--batch M]
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 5033.0      # 256 CU x 8192 int8 op/clk/CU x 2.4 GHz (dense; MI355X_MICROARCH.md: i8 = 2x the 2.5 PF bf16 rate)
M, K, N = 512, 4096, 11008   # BASELINE.json metric shape (Llama-2-7b up/gate projection, batch x seq = 512)
OUTLIER_FRAC = 0.01
SIGMA = 6
CLOCK_SETTLE_MS = 40.0       # untimed graph replays in front of the timed region: the chip's clock needs this long to settle under load


def timed_replays(graph, stream, replays, restore=None):
    """`replays` back-to-back replays of a captured hipGraph, every one between its own pair of HIP events on the launch stream, all
    queued before the host waits once: the device goes from one replay straight into the next (`restore()`, when given, puts the inputs
    back between two replays, OUTSIDE every event pair - the forward zeroes outlier columns in place).  Returns the list of per-replay
    milliseconds.  A single replay of a 20-step graph is a 0.6 ms sample with >= 7 % run-internal scatter; the median of
    >= 31 such samples is what bench.py reports."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(replays)]
    with torch.cuda.stream(stream):                                 # restore, events and replays all ordered on the launch stream
        for a, b in ev:
            if restore is not None:
                restore()
            a.record(stream)
            graph.replay()
            b.record(stream)
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


def median(vals):
    v = sorted(vals)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def conditioned_replay(graph, stream, restore=None, settle_ms=None, replays=9):
    """THE measurement protocol of this repository for a captured hipGraph (bench.py and every tools/ script that quotes microseconds):
    (1) the first replay after idle is timed and reported, never used; (2) the graph is replayed untimed for ~CLOCK_SETTLE_MS so the
    chip sits at the clock it sustains under this load (15-25 ms on MI355X; a 100-step graph is over in 3 ms); (3) `replays` replays
    back to back, each between its own event pair on the launch stream, the inputs restored between them (timed_replays); the MEDIAN
    is the figure.  Returns (median_ms, first_replay_ms, untimed_replays)."""
    settle_ms = CLOCK_SETTLE_MS if settle_ms is None else settle_ms
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record(stream)
    graph.replay()
    f1.record(stream)
    torch.cuda.synchronize()
    first_ms = f0.elapsed_time(f1)
    reps = min(4000, max(1, int(settle_ms / max(first_ms, 1e-3))))
    for _ in range(reps):
        graph.replay()
    torch.cuda.synchronize()
    return median(timed_replays(graph, stream, replays, restore)), first_ms, reps


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="forwards per timed region = per hipGraph (the driver runs --steps 20 --warmup 5)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--replays", type=int, default=31, help="back-to-back timed replays of the K-step graph; the median is reported")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--shape", default=None, help="K,N of the Linear (default 4096,11008); e.g. 8192,28672 for BASELINE config 3")
    ap.add_argument("--batch", type=int, default=None, help="token rows M (default 512)")
    ap.add_argument("--bit", type=int, choices=[8, 4], default=8, help="8: W8A8O16 (the metric); 4: W4A4O16, 128 static fp16 columns (BASELINE config 2)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the eager / cold-weights secondary timings")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly from Python instead of one hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: exercise the launch / rendezvous / gather path with synthetic per-rank timings (CPU, gloo)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# distributed helpers (exercised on CPU with gloo by tests/test_dist_gloo.py)
# ---------------------------------------------------------------------------------------------------------------
def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, LOCAL_RANK -> device), rendezvous
    on 127.0.0.1, pass rank 0's JSON line through, fail if any rank fails."""
    port = free_port()
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MIXQ_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:                                              # every child is reaped; the first failure stops the others
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:
                    q.terminate()
        if pending:
            time.sleep(0.05)
    return rc


def init_dist(n_gpus, backend):
    rank, local_rank, world = dist_env()
    if n_gpus > 1:
        if world != n_gpus:
            raise SystemExit(f"--gpus {n_gpus} needs WORLD_SIZE={n_gpus} (launch with python -m torch.distributed.run "
                             f"--nnodes=1 --nproc-per-node {n_gpus} --master-addr 127.0.0.1 ...); got WORLD_SIZE={world}")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _group_up():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def barrier(world, device=None):
    if world > 1 or _group_up():                          # (a 1-rank group exists only in the RCCL smoke test)
        import torch.distributed as dist
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def gather_counters(elapsed_s, flops, world, device, per_rank=False):
    """All ranks contribute {elapsed seconds, FLOPs}; returns (max elapsed, total FLOPs[, per-rank elapsed list]).  The only
    collective of the whole job (a few floats over xGMI); the data path itself shards by batch and exchanges nothing."""
    if world == 1 and not _group_up():
        return (elapsed_s, flops, [elapsed_s]) if per_rank else (elapsed_s, flops)
    import torch.distributed as dist
    mine = torch.tensor([elapsed_s, flops], dtype=torch.float64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    allv = torch.stack(allv).cpu()
    out = (float(allv[:, 0].max()), float(allv[:, 1].sum()))
    return out + ([float(v) for v in allv[:, 0]],) if per_rank else out


def describe_group(world, device, backend_arg):
    """What the process group itself reports - the backend, the number of ranks it sees, every rank's device - so that the `--gpus N` line
    answers "did RCCL see N ranks, one per GPU" by itself.  One all_gather_object of a short string per rank."""
    name = torch.cuda.get_device_name(device) + f" (cuda:{device.index})" if device.type == "cuda" else "cpu"
    if world == 1 and not _group_up():
        return {"backend": None, "world_size": 1, "devices": [name], "collectives": "none (single process)"}
    import torch.distributed as dist
    names = [None] * dist.get_world_size()
    dist.all_gather_object(names, name)
    return {"backend": dist.get_backend() + (" (RCCL on ROCm)" if dist.get_backend() == "nccl" else ""), "backend_requested": backend_arg,
            "world_size": dist.get_world_size(), "devices": names,
            "collectives": "barrier on both sides of the timed region, one all_gather of {elapsed, FLOPs} behind it; none on the data path"}


def shard_rows(total_rows, world, rank):
    """Row range of `rank` when a fixed global batch is split (strong scaling; token rows are independent)."""
    base, rem = divmod(total_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_rows(total_rows, world, rank, scaling):
    return shard_rows(total_rows, world, rank) if scaling == "strong" else (0, total_rows)


# ---------------------------------------------------------------------------------------------------------------
def outlier_columns(Kc=None):
    Kc = K if Kc is None else Kc
    g = torch.Generator().manual_seed(1)
    return torch.randperm(Kc, generator=g)[: round(OUTLIER_FRAC * Kc)]


def build_layer(device, rows, seed=0, bit=8, cache=None, shape=None):
    from mixq_amd import MixLibCache, MixLinear_GEMM
    K, N = shape if shape is not None else (globals()["K"], globals()["N"])
    torch.manual_seed(seed)
    lin = torch.nn.Linear(K, N, bias=False).half()            # nn.Linear default init, as examples/benchbitsand.py:519
    if cache is None:
        cache = MixLibCache(rows, sigma=SIGMA, bit=bit, device=device)
    if bit == 8:
        layer = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=device, name="up_proj")
    else:
        # W4A4O16 (linear.py:123-143): the 128 input channels with the largest calibration scale stay fp16; the synthetic
        # calibration marks the batch's outlier columns (and fills up to 128 with the next channels)
        scales = torch.ones(K) + torch.arange(K) * 1e-6
        cols = outlier_columns(K)
        scales[cols] = 20.0 + torch.arange(cols.numel()) * 1e-3
        layer = MixLinear_GEMM.from_linear(lin, 4, cache=cache, layer_scales=scales, dev=device, name="up_proj")
    return lin, cache, layer


def make_batches(count, rows, device, rank, Kc=None):
    Kc = K if Kc is None else Kc
    cols = outlier_columns(Kc)
    gx = torch.Generator().manual_seed(100 + rank)
    base = torch.randn(rows, Kc, generator=gx).half()
    base[:, cols] *= 20
    base = base.to(device)
    budget = 6 << 30                                                      # bytes of pristine inputs kept resident
    count = max(1, min(count, budget // max(1, base.numel() * 2)))
    pristine = base.unsqueeze(0).repeat(count, 1, 1).contiguous()        # [count, rows, K], one buffer per step (cycled if fewer)
    return cols, base, pristine


def cpu_baseline(rows):
    """The reference's CPU path for this metric (BASELINE.json config 0 / north_star): fp16 torch.nn.Linear on the host cores
    of this box, same shape, same init.  The thread count is swept (all logical CPUs oversubscribe the fp16 GEMM badly on
    SMT hosts) and the best is reported with its count; bounded to ~25 s in total."""
    logical = os.cpu_count() or 1
    torch.manual_seed(0)
    lin = torch.nn.Linear(K, N, bias=False).half()
    x = torch.randn(rows, K).half()
    cands = sorted({c for c in (logical, logical // 2, logical // 4, 32, 16, 8) if 1 <= c <= logical}, reverse=True)
    results = {}
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            for _ in range(2):
                lin(x)
            iters, t0 = 0, time.perf_counter()
            while iters < 50 and (iters < 3 or time.perf_counter() - t0 < 3.0):
                lin(x)
                iters += 1
            results[c] = (2.0 * rows * N * K * iters / (time.perf_counter() - t0) / 1e12, iters)
    best = max(results, key=lambda c: results[c][0])
    return {"value": round(results[best][0], 4), "unit": "TFLOPS", "cores": best, "kind": "reference",
            "sample": f"torch.nn.Linear({K},{N}).half() on CPU, M={rows}, {results[best][1]} forwards after 2 warm-ups at {best} threads "
                      f"(sweep over threads: " + ", ".join(f"{c}: {results[c][0]:.3f}" for c in cands) + f" TFLOPS; {logical} logical CPUs)"}


# BASELINE.json configs 2-4 at M = 512, timed by the default run AFTER the headline's timed region: the driver's one command
# then carries a driver-timed figure for every configuration, not only config 1's shape
SECONDARY_CONFIGS = [
    ("cfg2_w4a4_4096x11008", "BASELINE config 2: Llama-2-7b up_proj W4A4O16 (128 static fp16 columns), batch 512", 4, 4096, 11008),
    ("cfg3_w8a8_8192x28672", "BASELINE config 3: Llama-2-70b up_proj W8A8O16, batch 512 per GPU, 1 % outlier columns", 8, 8192, 28672),
    ("cfg4_w8a8_4096x14336", "BASELINE config 4: Llama-3-8B up_proj W8A8O16, batch 512, 1 % outlier columns, prediction on (frozen after 2 forwards)", 8, 4096, 14336),
]


def measure_config(device, side, rows, bit, Kc, Nc, steps=20, replays=15):
    """One BASELINE configuration under THE protocol of this file (conditioned_replay): the full forward (quantise + GEMM, a hipGraph of `steps`
    steps on pristine inputs, median of `replays` back-to-back replays) and the GEMM + fused epilogue alone.  Returns the dict that goes into
    `secondary_configs`."""
    from mixq_amd import _capi, mixlib
    _, cache, layer = build_layer(device, rows, seed=7, bit=bit, shape=(Kc, Nc))
    _, base, pristine = make_batches(steps, rows, device, 0, Kc)
    nbuf = pristine.shape[0]
    for _ in range(3):
        layer(base.clone(), None, True)
    torch.cuda.synchronize()
    assert layer.add_outliers is False
    n_ind = int(layer.ind.numel())
    restore = lambda: pristine.copy_(base.unsqueeze(0).expand_as(pristine))
    with torch.cuda.stream(side):
        for i in range(3):
            layer(pristine[i % nbuf], None, True)
        torch.cuda.synchronize()
        restore()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(steps):
                layer(pristine[i % nbuf], None, True)
        torch.cuda.synchronize()
        step_ms, _, _ = conditioned_replay(g, side, restore=restore, replays=replays)
        restore()
        ind_buf, n_dev = layer._ind_dev()
        q_x, x_out = mixlib.QuantFused(base.clone(), ind_buf, cache.x_scale, bit, SIGMA, n_dev=n_dev, fmt=layer.x_fmt())
        cache.q_xcache, cache.activation_outliers, cache.n_dev = q_x, (x_out[:, :n_ind] if n_ind else None), n_dev
        for _ in range(3):
            layer._gemm(cache, rows, 0)
        torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg, stream=side):
            for _ in range(steps):
                layer._gemm(cache, rows, 0)
        torch.cuda.synchronize()
        gemm_ms, _, _ = conditioned_replay(gg, side, replays=replays)
    flops = 2.0 * rows * Nc * Kc
    us, gus = step_ms * 1e3 / steps, gemm_ms * 1e3 / steps
    f6 = layer.x_fmt() == 4
    peak = PEAK_INT8_TOPS * (2 if f6 else 1)
    cfg = _capi.gemm_config_names()[_capi.load().mixq_gemm_pick_config_fmt(rows, Nc, Kc, bit, mixlib.fmt_of(layer._wpk))]
    out = {"bit": bit, "M": rows, "K": Kc, "N": Nc, "outlier_columns": n_ind, "ms_per_step": round(us * 1e-3, 5), "tflops": round(flops / us / 1e6, 1),
           "pct_of_int8_mfma_peak": round(100.0 * flops / us / 1e6 / PEAK_INT8_TOPS, 2),
           "gemm": {"kernel": ("FP6-pipe MFMA GEMM" if f6 else "int8 MFMA GEMM") + f" + fused epilogue ({cfg})", "us_per_launch": round(gus, 3),
                    "achieved_tflops": round(flops / gus / 1e6, 1), "peak": peak, "frac": round(flops / gus / 1e6 / peak, 4)}}
    del g, gg, layer, pristine
    torch.cuda.empty_cache()
    return out


def hbm_traffic_from_profile(tag="", cfgname=None):
    """(HBM bytes per GEMM launch, source file) from the committed rocprofv3 PMC passes - the newest profiles/rNN_hbm_traffic<tag>.json -
    or (None, reason).  The counters need their own rocprofv3 --pmc passes (tools/r05_profile.sh), so the bench line carries the committed
    figure of the same configuration (tag "" = the metric shape W8A8, "_w4a4", "_8192x28672") - and ONLY while the kernel it was
    measured on is the kernel this run picked: the file names the kernel instantiation (`gemm_wreg_kernel<MB, WNB, ...>`), the tiling's name
    (`wr<16 MB>x<64 WNB>_...`) must agree, otherwise the figure is stale and the line says so instead of carrying it."""
    try:
        best = src = None
        pdir = os.path.join(ROOT, "profiles")
        for f in sorted(os.listdir(pdir)):
            if f.endswith(f"hbm_traffic{tag}.json"):
                best, src = json.load(open(os.path.join(pdir, f))), "profiles/" + f
        if best is None:
            return None, None
        if cfgname:
            import re
            m, k = re.match(r"wr(\d+)x(\d+)_", cfgname), re.search(r"gemm_wreg_kernel<(\d+), (\d+),", best.get("kernel", ""))
            if not m or not k or (int(m.group(1)) // 16, int(m.group(2)) // 64) != (int(k.group(1)), int(k.group(2))):
                return None, f"stale: {src} was measured on {best.get('kernel', '?')[:60]}, this run picked {cfgname}"
        return best.get("hbm_bytes_per_launch"), src
    except Exception:
        return None, None


def dry_run(args):
    """CPU rehearsal of the N-rank job: rendezvous, row sharding, barriers, the counter gather and the JSON line, with a
    synthetic per-rank time instead of GPU work (tests/test_dist_gloo.py drives `--gpus 2 --backend gloo --dry-run`)."""
    rank, local_rank, world = init_dist(args.gpus, args.backend)
    dev = torch.device("cpu")
    lo, hi = rank_rows(M, world, rank, args.scaling)
    barrier(world, dev)
    elapsed = 1e-3 * (1.0 + 0.1 * rank)                                   # rank r "takes" 1 + 0.1 r ms for its rows
    flops = 2.0 * (hi - lo) * N * K * args.steps
    barrier(world, dev)
    mx, tot, per = gather_counters(elapsed, flops, world, dev, per_rank=True)
    dist_info = describe_group(world, dev, args.backend)
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "dry_run": True, "distributed": dist_info, "value": round(tot / mx / 1e12, 3), "unit": "TFLOPS",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling,
                          "rows_per_rank": hi - lo, "per_rank_ms": [round(v * 1e3, 4) for v in per],
                          "config": {"workload": "launch-path rehearsal", "M": M, "K": K, "N": N}}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


def main(argv=None):
    global M, K, N
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.shape:
        K, N = (int(v) for v in args.shape.split(","))
    if args.batch:
        M = args.batch
    if args.gpus > 1 and dist_env()[2] != args.gpus and not os.environ.get("MIXQ_BENCH_CHILD"):
        return spawn_ranks(args.gpus, argv)                             # no launcher around us: be the launcher
    if args.dry_run:
        return dry_run(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: mixq_amd has no CPU path (the CPU numbers it prints are the baseline only)")
    torch.cuda.set_device(dist_env()[1])                         # before the process group exists: RCCL binds to the current device
    rank, local_rank, world = init_dist(args.gpus, args.backend)
    device = torch.device("cuda", local_rank)
    from mixq_amd import _capi, mixlib
    info = _capi.device_info()                                  # loads libmixq_hip.so; raises if it is missing

    lo, hi = rank_rows(M, world, rank, args.scaling)
    rows = hi - lo
    bit = args.bit
    lin, cache, layer = build_layer(device, rows, bit=bit)
    steps, warm = args.steps, args.warmup
    cols, base, pristine = make_batches(steps, rows, device, rank)
    nbuf = pristine.shape[0]

    # outlier prediction warm-up: the first cache.stop (=2) forwards discover and freeze `ind` (host syncs allowed here)
    for _ in range(3):
        xw = base.clone()
        y = layer(xw, None, True)
    torch.cuda.synchronize()
    assert layer.add_outliers is False
    n_ind = int(layer.ind.numel())

    # quick self-check against a Linear over the dequantised operands on the GPU (the CPU-oracle parity lives in tests/)
    with torch.no_grad():
        xz = base.clone()
        q, xo = mixlib.QuantFused(xz, layer.ind, cache.x_scale, bit, SIGMA)
        if bit == 4:                                            # nibbles -> signed values, even column in the low nibble
            lo, hi = (q & 15).to(torch.int8), (q >> 4).to(torch.int8)
            q = torch.stack((torch.where(lo > 7, lo - 16, lo), torch.where(hi > 7, hi - 16, hi)), dim=2).reshape(rows, K)
            qw = layer.q_weight
            wl, wh = (qw & 15).to(torch.int8), (qw >> 4).to(torch.int8)
            Wq = torch.stack((torch.where(wl > 7, wl - 16, wl), torch.where(wh > 7, wh - 16, wh)), dim=2).reshape(N, K)
        else:
            Wq = layer.q_weight
        Xd = q.double() * cache.x_scale[:rows].double()
        Wd = Wq.double() * layer.scale_col.double().T
        Wd[:, layer.ind.long()] = layer.weight_cache.double()    # the fp16 columns the operator actually multiplies (linear.py:207)
        Xd[:, layer.ind.long()] = base[:, layer.ind.long()].double()
        ref = Xd @ Wd.T
        max_abs_err = float((layer(base.clone(), None, True).double() - ref).abs().max())
        del Xd, Wd, ref, Wq

    def one_step(i):
        return layer(pristine[i % nbuf], None, True)

    side = torch.cuda.Stream(device=device)
    graph = None
    first_replay_ms, settle_reps = None, 0
    with torch.cuda.stream(side):
        for i in range(min(warm, steps)):                       # untimed warm-up steps (on buffers restored below)
            one_step(i)
        torch.cuda.synchronize()
        pristine.copy_(base.unsqueeze(0).expand_as(pristine))
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for i in range(steps):
                    one_step(i)
            torch.cuda.synchronize()
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(side)
            graph.replay()                                      # first replay after idle: graph upload, cold clocks (reported, not `value`)
            f1.record(side)
            torch.cuda.synchronize()
            first_replay_ms = f0.elapsed_time(f1)
            settle_reps = min(4000, max(1, int(CLOCK_SETTLE_MS / max(first_replay_ms, 1e-3))))
            for _ in range(settle_reps):                        # untimed: bring the chip to the clock it sustains under this load
                graph.replay()
            torch.cuda.synchronize()
            pristine.copy_(base.unsqueeze(0).expand_as(pristine))
        restore_inputs = lambda: pristine.copy_(base.unsqueeze(0).expand_as(pristine))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        barrier(world, device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        replay_ms = None
        if graph is not None:
            # the K-step graph replayed `--replays` times back to back, each replay between its own pair of events (the device never
            # idles between them; the inputs are restored between replays, outside the event pairs): the MEDIAN replay is the figure
            replay_ms = timed_replays(graph, side, args.replays, restore_inputs)
            host_elapsed = (time.perf_counter() - t0) / args.replays
            elapsed = median(replay_ms) * 1e-3
        else:
            e0.record(side)
            for i in range(steps):
                one_step(i)
            e1.record(side)
            torch.cuda.synchronize()
            host_elapsed = time.perf_counter() - t0
            elapsed = e0.elapsed_time(e1) * 1e-3                # seconds between the two events on the launch stream
        barrier(world, device)
        # the previous rounds' figure, kept beside the median: ONE replay between two events, from a synchronised (idle) stream
        single_ms = None
        if graph is not None:
            restore_inputs()
            torch.cuda.synchronize()
            e0.record(side)
            graph.replay()
            e1.record(side)
            torch.cuda.synchronize()
            single_ms = e0.elapsed_time(e1)

        # ---- dominant kernel (the int8 MFMA GEMM + fused epilogue) alone, HIP events on the launch stream ------------
        ind_buf, n_dev = layer._ind_dev()
        q_x, x_out = mixlib.QuantFused(base.clone(), ind_buf, cache.x_scale, bit, SIGMA, n_dev=n_dev, fmt=layer.x_fmt())
        cache.q_xcache, cache.activation_outliers, cache.n_dev = q_x, (x_out[:, :n_ind] if n_ind else None), n_dev
        gsteps = max(20, min(steps, 200))
        for _ in range(5):
            layer._gemm(cache, rows, 0)
        torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg, stream=side):
            for _ in range(gsteps):
                layer._gemm(cache, rows, 0)
        torch.cuda.synchronize()
        gemm_ms, gemm_first_ms, _ = conditioned_replay(gg, side, settle_ms=0.0 if args.no_graph else None)   # the same clock conditioning as for the step
        gemm_us = gemm_ms * 1e3 / gsteps

        # ---- secondary timings, outside the timed region of `value` --------------------------------------
        eager_ms = cold_ms = None
        cold_note = None
        mlp_ms = {}
        secondary = {}
        if not args.no_secondary and rank == 0:                 # (rank 0 only: the other ranks wait for it in the counter gather)
            # (a) the reference's own protocol, examples/benchbitsand.py:534-550: NO graph, 10 warm-up + 100 timed back-to-back
            # `layer(x)` calls from Python between two events - what a plain Hugging Face loop pays per layer, host cost included
            esteps = 100
            pristine.copy_(base.unsqueeze(0).expand_as(pristine))
            for i in range(10):
                one_step(i)
            torch.cuda.synchronize()
            pristine.copy_(base.unsqueeze(0).expand_as(pristine))
            torch.cuda.synchronize()
            h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h0.record(side)
            for i in range(esteps):
                one_step(i)
            h1.record(side)
            torch.cuda.synchronize()
            eager_ms = h0.elapsed_time(h1) / esteps
            # (b) cold weights: one graph whose steps rotate through enough copies of the layer to exceed the 256 MB MALL, as the
            # layers of a model do (the headline loop, like the reference's, re-uses ONE layer whose weights stay cache-resident)
            wbytes = int(layer._wpk.numel())
            copies = max(8, -(-300 * (1 << 20) // wbytes) + 1)
            layers = [layer]
            for c in range(1, copies):
                _, _, lc = build_layer(device, rows, seed=c, bit=bit, cache=cache)
                for _ in range(3):
                    lc(base.clone(), None, True)
                layers.append(lc)
            torch.cuda.synchronize()
            csteps = max(copies * 4, 40)
            pristine.copy_(base.unsqueeze(0).expand_as(pristine))
            for i in range(copies):
                layers[i](pristine[i % nbuf], None, True)
            torch.cuda.synchronize()
            pristine.copy_(base.unsqueeze(0).expand_as(pristine))
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg, stream=side):
                for i in range(csteps):
                    layers[i % copies](pristine[i % nbuf], None, True)
            torch.cuda.synchronize()
            cg.replay()
            torch.cuda.synchronize()
            pristine.copy_(base.unsqueeze(0).expand_as(pristine))
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(side)
            cg.replay()
            c1.record(side)
            torch.cuda.synchronize()
            cold_ms = c0.elapsed_time(c1) / csteps
            cold_note = f"one hipGraph of {csteps} steps rotating over {copies} layer copies ({copies * wbytes >> 20} MB of weights > 256 MB MALL)"
            del layers, cg
            # (c) the MLP block this layer sits in (mixquant/modules/fused/mlp.py:57-70: fused norm -> up_proj, gate_proj -> down_proj) at the
            # same batch: gate_proj + up_proj as ONE launch over their interleaved rows (MIXQ_ACT_SILU_PAIR) against the two launches
            try:
                from mixq_amd import FasterTransformerRMSNorm, MixLibCache, MixLinear_GEMM, MixLlamaMLP
                bcache = MixLibCache(rows, sigma=SIGMA, bit=bit, device=device)
                _, _, up_l = build_layer(device, rows, seed=21, bit=bit, cache=bcache)
                _, _, gate_l = build_layer(device, rows, seed=22, bit=bit, cache=bcache)
                torch.manual_seed(23)
                down_l = MixLinear_GEMM.from_linear(torch.nn.Linear(N, K, bias=False).half(), 8, cache=bcache, dev=device, name="down_proj")
                norm = FasterTransformerRMSNorm((torch.rand(K) + 0.5).half().to(device), 1e-5, bcache)
                norm.next_layer = up_l
                block = MixLlamaMLP(gate_l, down_l, up_l, bcache)
                bsteps = 20
                bx = base.unsqueeze(0).repeat(bsteps, 1, 1).contiguous()
                for joint in (False, True):
                    block.config.joint_gate_up = joint                  # (bcache's MixqConfig: this block's switches only)
                    for _ in range(4):
                        block(norm(base.clone()))                   # freezes the predictions; the joint image is built on the first frozen forward
                    torch.cuda.synchronize()
                    bg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(bg, stream=side):
                        for i in range(bsteps):
                            block(norm(bx[i]))
                    torch.cuda.synchronize()
                    bms, _, _ = conditioned_replay(bg, side, restore=lambda: bx.copy_(base.unsqueeze(0).expand_as(bx)))
                    mlp_ms[joint] = bms / bsteps
                    del bg
                del block, up_l, gate_l, down_l, bx
            except Exception as exc:                                # (a secondary figure must not take the bench line down)
                print(f"bench.py: MLP-block secondary timing skipped: {exc!r}", file=sys.stderr)
            # (d) BASELINE configs 2-4 at their own shapes, same protocol (only from the metric's own invocation: a --shape / --bit / --batch run
            # is somebody measuring ONE configuration)
            if not args.shape and not args.batch and bit == 8 and world == 1:
                for tag, what, sbit, sk, sn in SECONDARY_CONFIGS:
                    try:
                        secondary[tag] = dict(measure_config(device, side, rows, sbit, sk, sn), what=what)
                    except Exception as exc:
                        print(f"bench.py: secondary config {tag} skipped: {exc!r}", file=sys.stderr)

    flops_step = 2.0 * rows * N * K
    max_elapsed, total_flops, per_rank = gather_counters(elapsed, flops_step * steps, world, device, per_rank=True)
    max_host, _ = gather_counters(host_elapsed, 0.0, world, device)
    value = total_flops / max_elapsed / 1e12
    ms_per_step = max_elapsed * 1e3 / steps
    dist_info = describe_group(world, device, args.backend)
    achieved = flops_step / (gemm_us * 1e-6) / 1e12

    if rank == 0:
        fmt = layer.x_fmt()
        # (the committed PMC passes profiled the metric configuration: any other shape / bit width / per-rank batch carries no traffic figure)
        ttag = {(8, 512, 4096, 11008): "", (4, 512, 4096, 11008): "_w4a4", (8, 512, 8192, 28672): "_8192x28672"}.get((bit, rows, K, N))
        cfg_now = _capi.gemm_config_names()[_capi.load().mixq_gemm_pick_config_fmt(rows, N, K, bit, mixlib.fmt_of(layer._wpk))]
        traffic, traffic_src = hbm_traffic_from_profile(ttag, cfg_now) if ttag is not None else (None, None)
        shape_note = "Llama-2-7b up_proj shape" if (K, N) == (4096, 11008) else f"{K}->{N}"
        out = {
            "metric": f"effective int8 TFLOPS, W{bit}A{bit}O16 MixQ Linear forward (quantise + {'int8' if bit == 8 or fmt != 4 else 'FP6-pipe'} MFMA GEMM + fused dequant/outlier "
                      f"epilogue), batch {M}, {K}->{N}",
            "value": round(value, 2), "unit": "TFLOPS", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "int8" if bit == 8 else ("int4 (carried exactly as FP6 E3M2 codes on the FP6 matrix pipe: CDNA4 has no int4 MFMA)" if fmt == 4
                                              else "int4 (expanded to int8 in registers: CDNA4 has no int4 MFMA)"), "data": "synthetic",
            "config": {"workload": f"MixLinear_GEMM W{bit}A{bit}O16 forward, {shape_note}", "M": M, "K": K, "N": N,
                       "outlier_columns": n_ind, "outlier_predict": "frozen after 2 warm-up forwards", "sigma": SIGMA,
                       "weights": "nn.Linear default init, quantised per output channel", "per_gpu_batch": rows,
                       "parallelism": f"batch-shard x{world} ({'independent replicas' if args.scaling == 'weak' else 'rows of one batch split'}, "
                                      f"no data-path collective)",
                       "launch": "eager" if args.no_graph else f"one hipGraph of {steps} steps",
                       "operand_format": {"activations": {0: "plain", 1: "P16x64", 4: "R6x128 (int4 as FP6 E3M2 codes, row-contiguous)"}[fmt],
                               "weights": {0: "plain", 1: "P16x64", 2: "F16x64", 3: "F6x128 (int4 as FP6 E3M2 codes)"}[mixlib.fmt_of(layer._wpk)]},
                       "weight_bytes_resident": int(layer._wpk.numel() + (0 if layer._buffers['q_weight'] is None else layer._buffers['q_weight'].numel()))},
            "timing": {"clock": "HIP events on the launch stream around the K steps" if graph is None else
                                f"HIP events on the launch stream around each of {args.replays} back-to-back replays of the K-step graph (inputs restored "
                                f"between replays, outside the event pairs); ms_per_step = MEDIAN replay / K",
                       "replays": None if replay_ms is None else len(replay_ms),
                       "replay_ms_min": None if replay_ms is None else round(min(replay_ms), 5),
                       "replay_ms_median": None if replay_ms is None else round(median(replay_ms), 5),
                       "replay_ms_mean": None if replay_ms is None else round(sum(replay_ms) / len(replay_ms), 5),
                       "replay_ms_max": None if replay_ms is None else round(max(replay_ms), 5),
                       "single_replay_ms_per_step": None if single_ms is None else round(single_ms / steps, 5),
                       "single_replay_protocol": "ONE replay between two events from a synchronised stream (rounds 1-4's figure; carries the graph-launch latency)",
                       "host_wall_ms_per_step": round(max_host * 1e3 / steps, 5),
                       "per_rank_ms_per_step": [round(v * 1e3 / steps, 5) for v in per_rank],
                       "clock_settle": None if graph is None else f"{settle_reps} untimed replays of the captured graph (~{CLOCK_SETTLE_MS:.0f} ms) between the warm-up and the timed region",
                       "first_replay_ms_per_step": None if first_replay_ms is None else round(first_replay_ms / steps, 5),
                       "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 5),
                       "eager_protocol": "no graph: 10 warm-up + 100 back-to-back layer(x) calls from Python between two events (examples/benchbitsand.py:534-550), rank 0",
                       "cold_weights_ms_per_step": None if cold_ms is None else round(cold_ms, 5), "cold_weights_protocol": cold_note,
                       "mlp_block_ms": None if True not in mlp_ms else round(mlp_ms[True], 5),
                       "mlp_block_two_launches_ms": None if False not in mlp_ms else round(mlp_ms[False], 5),
                       "mlp_block_protocol": "fused RMSNorm + quantise -> gate_proj + up_proj (one launch over interleaved rows | two launches) -> down_proj with the row "
                                             "maxima from the producer, same batch and shape (hidden -> intermediate -> hidden), graph of 20 blocks, same clock conditioning"},
            "pct_of_int8_mfma_peak": round(100.0 * value / (PEAK_INT8_TOPS * world), 2),
            # rounds 1-4 reported ONE replay between two events from an idle stream: the same figure of THIS run, for comparison across rounds
            "ms_per_step_single_replay": None if single_ms is None else round(single_ms / steps, 5),
            "tflops_single_replay": None if single_ms is None else round(flops_step * world * steps / (single_ms * 1e-3) / 1e12, 2),
            "secondary_configs": secondary or None,
            "secondary_protocol": "after the headline's timed region, rank 0, same conditioned_replay protocol: hipGraph of 20 forwards on pristine inputs, "
                                  "first replay discarded, ~40 ms of untimed replays, median of 15 back-to-back replays; `gemm` = GEMM + fused epilogue alone",
            "distributed": dist_info,
            "max_abs_err_vs_dequant_linear": round(max_abs_err, 6),
            # (W4A4 on the FP6 pipe is priced against the FP6 dense peak, 2x the int8 one: MI355X_MICROARCH.md "Peak FP6/FP4 MFMA ~10 PF dense")
            "roofline": {"bound": "mfma", "kernel": ("FP6-pipe MFMA GEMM (int4 as E3M2 codes)" if fmt == 4 else "int8 MFMA GEMM") + " + fused epilogue (" +
                                                    _capi.gemm_config_names()[_capi.load().mixq_gemm_pick_config_fmt(rows, N, K, bit, mixlib.fmt_of(layer._wpk))] + ")",
                         "rank": 0,
                         "achieved": round(achieved, 2), "peak": PEAK_INT8_TOPS * (2 if fmt == 4 else 1), "unit": "TFLOP/s",
                         "frac": round(achieved / (PEAK_INT8_TOPS * (2 if fmt == 4 else 1)), 4), "traffic": traffic, "traffic_source": traffic_src,
                         "us_per_launch": round(gemm_us, 3), "algorithmic_flops_per_launch": flops_step,
                         "algorithmic_bytes_per_launch": (rows * K + N * K) * bit // 8 + 2 * rows * N},
            "device": info,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rows)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
