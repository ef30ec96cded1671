#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

metric  : effective int8 TFLOPS (= 2*M*N*K / t, outlier FLOPs and quantise work not counted, the reference's
          convention examples/benchbitsand.py:554-555) of the W8A8O16 MixQ Linear, (batch x seq, hidden) = (512, 4096)
          -> 11008, with 1 % synthetic outlier columns and outlier prediction frozen after 2 warm-up forwards.
step    : ONE full forward of MixLinear_GEMM over one 512-token batch already resident in HBM:
          (i)+(ii) fused extract/zero/absmax/quantise kernel  +  (iii)+(iv) int8 MFMA GEMM with fused dequant,
          fp16 outlier tail and (absent for Llama) bias.  Each step gets its own pristine activation buffer (the
          operator zeroes outlier columns in place, as the reference does), so no restore copy sits in the timed region.
N GPUs  : weak scaling, one process per GPU, every rank runs the same K steps on its own batches; no data-path
          collective, one RCCL all_gather of {elapsed, flops} at the end.  value = sum of FLOPs over ranks / max time.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]     (N > 1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 5033.0      # 256 CU x 8192 int8 op/clk/CU x 2.4 GHz (dense; MI355X_MICROARCH.md: i8 = 2x the 2.5 PF bf16 rate)
M, K, N = 512, 4096, 11008   # BASELINE.json metric shape (Llama-2-7b up/gate projection, batch x seq = 512)
OUTLIER_FRAC = 0.01
SIGMA = 6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly from Python instead of one hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# distributed helpers (exercised on CPU with gloo by tests/test_dist_gloo.py)
# ---------------------------------------------------------------------------------------------------------------
def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


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


def barrier(world, device=None):
    if world > 1:
        import torch.distributed as dist
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def gather_counters(elapsed_s, flops, world, device):
    """All ranks contribute {elapsed seconds, FLOPs}; returns (max elapsed, total FLOPs).  The only collective of the
    whole job (a few floats over xGMI); the data path itself shards by batch and exchanges nothing."""
    if world == 1:
        return elapsed_s, flops
    import torch.distributed as dist
    mine = torch.tensor([elapsed_s, flops], dtype=torch.float64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    allv = torch.stack(allv).cpu()
    return float(allv[:, 0].max()), float(allv[:, 1].sum())


def shard_rows(total_rows, world, rank):
    """Row range of `rank` when a fixed global batch is split (strong scaling helper; rows are independent)."""
    base, rem = divmod(total_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------------------
def build_layer(device, seed=0):
    from mixq_amd import MixLibCache, MixLinear_GEMM
    torch.manual_seed(seed)
    lin = torch.nn.Linear(K, N, bias=False).half()            # nn.Linear default init, as examples/benchbitsand.py:519
    cache = MixLibCache(M, sigma=SIGMA, bit=8, device=device)
    layer = MixLinear_GEMM.from_linear(lin, 8, cache=cache, dev=device, name="up_proj")
    return lin, cache, layer


def make_batches(count, device, rank):
    g = torch.Generator().manual_seed(1)
    cols = torch.randperm(K, generator=g)[: round(OUTLIER_FRAC * K)]
    gx = torch.Generator().manual_seed(100 + rank)
    base = torch.randn(M, K, generator=gx).half()
    base[:, cols] *= 20
    base = base.to(device)
    pristine = base.unsqueeze(0).repeat(count, 1, 1).contiguous()      # [count, M, K], one buffer per step
    return cols, base, pristine


def cpu_baseline():
    """The reference's CPU path for this metric (BASELINE.json config 0 / north_star): fp16 torch.nn.Linear on the
    host cores of this box, same shape, same init; bounded to ~10 s."""
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    lin = torch.nn.Linear(K, N, bias=False).half()
    x = torch.randn(M, K).half()
    with torch.no_grad():
        for _ in range(2):
            lin(x)
        iters, t0 = 0, time.perf_counter()
        while iters < 200 and (iters < 5 or time.perf_counter() - t0 < 10.0):
            lin(x)
            iters += 1
        dt = time.perf_counter() - t0
    tflops = 2.0 * M * N * K * iters / dt / 1e12
    return {"value": round(tflops, 4), "unit": "TFLOPS", "cores": cores, "kind": "reference",
            "sample": f"torch.nn.Linear({K},{N}).half() on CPU, M={M}, {iters} forwards after 2 warm-ups, {dt:.1f} s, "
                      f"{torch.get_num_threads()} threads"}


def hbm_traffic_from_profile():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (profiles/*hbm_traffic.json), or None."""
    try:
        best = None
        pdir = os.path.join(ROOT, "profiles")
        for f in sorted(os.listdir(pdir)):
            if f.endswith("hbm_traffic.json"):
                best = json.load(open(os.path.join(pdir, f)))
        return None if best is None else best.get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse()
    rank, local_rank, world = init_dist(args.gpus, args.backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: mixq_amd has no CPU path (the CPU numbers it prints are the baseline only)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from mixq_amd import _capi, mixlib
    info = _capi.device_info()                                  # loads libmixq_hip.so; raises if it is missing

    lin, cache, layer = build_layer(device)
    steps, warm = args.steps, args.warmup
    cols, base, pristine = make_batches(steps, device, rank)

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
        q, xo = mixlib.QuantFused(xz, layer.ind, cache.x_scale, 8, SIGMA)
        Xd = q.double() * cache.x_scale[:M].double()
        Xd[:, layer.ind.long()] = base[:, layer.ind.long()].double()
        ref = Xd @ (layer.q_weight.double() * layer.scale_col.double().T).T
        max_abs_err = float((layer(base.clone(), None, True).double() - ref).abs().max())

    def one_step(i):
        return layer(pristine[i], None, True)

    side = torch.cuda.Stream(device=device)
    graph = None
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
            graph.replay()                                      # one untimed replay: graph upload, clocks
            torch.cuda.synchronize()
            pristine.copy_(base.unsqueeze(0).expand_as(pristine))
        torch.cuda.synchronize()
        barrier(world, device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            for i in range(steps):
                one_step(i)
        torch.cuda.synchronize()
        barrier(world, device)
        elapsed = time.perf_counter() - t0

        # ---- dominant kernel (the int8 MFMA GEMM + fused epilogue) alone, HIP events on the launch stream ------------
        q_x, x_out = mixlib.QuantFused(base.clone(), layer.ind, cache.x_scale, 8, SIGMA, packed=True)
        cache.q_xcache, cache.q_xcache_packed, cache.activation_outliers = q_x, True, x_out
        gsteps = max(20, min(steps, 200))
        for _ in range(5):
            layer._gemm(cache, M, 0)
        torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg, stream=side):
            for _ in range(gsteps):
                layer._gemm(cache, M, 0)
        torch.cuda.synchronize()
        gg.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        gg.replay()
        e1.record(side)
        torch.cuda.synchronize()
        gemm_us = e0.elapsed_time(e1) * 1e3 / gsteps

    flops_step = 2.0 * M * N * K
    max_elapsed, total_flops = gather_counters(elapsed, flops_step * steps, world, device)
    value = total_flops / max_elapsed / 1e12
    ms_per_step = max_elapsed * 1e3 / steps
    achieved = flops_step / (gemm_us * 1e-6) / 1e12

    if rank == 0:
        out = {
            "metric": "effective int8 TFLOPS, W8A8O16 MixQ Linear forward (quantise + int8 MFMA GEMM + fused dequant/outlier "
                      "epilogue), batch 512, 4096->11008",
            "value": round(value, 2), "unit": "TFLOPS", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8", "data": "synthetic",
            "config": {"workload": "MixLinear_GEMM W8A8O16 forward, Llama-2-7b up_proj shape", "M": M, "K": K, "N": N,
                       "outlier_columns": n_ind, "outlier_predict": "frozen after 2 warm-up forwards", "sigma": SIGMA,
                       "weights": "nn.Linear default init, quantised per output channel", "per_gpu_batch": M,
                       "parallelism": f"batch-shard x{world} (independent replicas, no data-path collective)",
                       "launch": "eager" if args.no_graph else f"one hipGraph of {steps} steps"},
            "pct_of_int8_mfma_peak": round(100.0 * value / (PEAK_INT8_TOPS * world), 2),
            "max_abs_err_vs_dequant_linear": round(max_abs_err, 6),
            "roofline": {"bound": "mfma", "kernel": "gemm_kernel (int8 MFMA GEMM + fused epilogue)",
                         "achieved": round(achieved, 2), "peak": PEAK_INT8_TOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_INT8_TOPS, 4), "traffic": hbm_traffic_from_profile(),
                         "us_per_launch": round(gemm_us, 3), "algorithmic_flops_per_launch": flops_step,
                         "algorithmic_bytes_per_launch": 2 * M * K // 2 + N * K + 2 * M * N,
                         "gemm_config": _capi.gemm_config_names()[_capi.load().mixq_gemm_pick_config(M, N, K, 8)]},
            "device": info,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
