"""The N > 1 path of bench.py on CPU: world_size 2 over gloo, rendezvous on 127.0.0.1.  The data path shards by batch
and exchanges nothing; the only collective is the all_gather of {elapsed, FLOPs} (bench.gather_counters)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = bench.init_dist(world, "gloo")
    assert (r, lr, w) == (rank, rank, world)
    dev = torch.device("cpu")
    bench.barrier(w, dev)
    # every rank "measures" a different time for the same work; the job time is the max, the work is the sum
    elapsed, flops = (0.5 if rank == 0 else 0.8), 100.0 * (rank + 1)
    mx, tot = bench.gather_counters(elapsed, flops, w, dev)
    assert bench.gather_counters(elapsed, flops, w, dev, per_rank=True)[2] == [0.5, 0.8]
    lo, hi = bench.shard_rows(513, w, rank)
    q.put((rank, mx, tot, lo, hi))
    bench.barrier(w, dev)
    import torch.distributed as dist
    dist.destroy_process_group()


def test_world2_gloo_gather_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mx, tot, lo, hi in res:
        assert mx == pytest.approx(0.8) and tot == pytest.approx(300.0)
    assert (res[0][3], res[0][4]) == (0, 257) and (res[1][3], res[1][4]) == (257, 513)     # disjoint cover of the rows


def test_shard_rows_partition():
    for total in (0, 1, 7, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [bench.shard_rows(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_counters_passthrough():
    assert bench.gather_counters(1.5, 7.0, 1, torch.device("cpu")) == (1.5, 7.0)


def _run_bench(*argv, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env or dict(os.environ), capture_output=True, text=True,
                          timeout=600)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launches_its_own_ranks_dry_run(scaling):
    """`python bench.py --gpus 2` with no launcher around it: bench.py spawns the 2 ranks itself (the driver's command form),
    they rendezvous over gloo on 127.0.0.1, gather their counters and rank 0 prints ONE JSON line.  --dry-run replaces the
    GPU work by a synthetic time per rank so the whole launch path runs on the CPU box."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MIXQ_BENCH_CHILD")}
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3", "--scaling", scaling, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["scaling"] == scaling
    # the line says what the process group itself saw: backend, world size, one device per rank
    d = out["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["devices"] == ["cpu", "cpu"] and "all_gather" in d["collectives"]
    assert out["per_rank_ms"] == [1.0, 1.1]                              # max over ranks is what the value is computed with
    rows = 256 if scaling == "strong" else 512
    assert out["rows_per_rank"] == rows
    total = 2.0 * 512 * 11008 * 4096 * 3 * (1 if scaling == "strong" else 2)
    assert out["value"] == pytest.approx(total / 1.1e-3 / 1e12, rel=1e-3)


def test_config3_scale_command_rehearsed_two_ranks_strong_70b_shape():
    """BASELINE config 3's exact SCALE command form (Llama-2-70b up_proj 8192 -> 28672, one 512-token batch split over the ranks) rehearsed
    on CPU: `bench.py --gpus 2 --shape 8192,28672 --scaling strong` self-launches its two ranks over gloo, each takes 256 rows of the ONE
    batch (no data-path collective: rows are independent), one all_gather of the counters, rank 0 prints the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MIXQ_BENCH_CHILD")}
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--shape", "8192,28672", "--scaling", "strong", "--steps", "4", env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 4 and out["dry_run"] is True
    assert out["config"]["K"] == 8192 and out["config"]["N"] == 28672 and out["config"]["M"] == 512
    assert out["rows_per_rank"] == 256 and out["per_rank_ms"] == [1.0, 1.1]
    total = 2.0 * 512 * 28672 * 8192 * 4                                  # ONE batch per step however many ranks share it
    assert out["value"] == pytest.approx(total / 1.1e-3 / 1e12, rel=1e-3)


def test_bench_dry_run_under_an_external_launcher_shape_flag():
    """The torch.distributed.run form: WORLD_SIZE already equals --gpus, so bench.py must NOT spawn again.  World size 1 here
    (a second level of processes is what the test above covers); --shape / --batch reach the job description."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = _run_bench("--gpus", "1", "--dry-run", "--shape", "8192,28672", "--batch", "64", "--steps", "2", env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["K"] == 8192 and out["config"]["N"] == 28672 and out["config"]["M"] == 64 and out["n_gpus"] == 1


def test_self_launched_ranks_fail_loudly_without_gpus():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MIXQ_BENCH_CHILD")}
    r = _run_bench("--gpus", "2", "--steps", "1", "--backend", "gloo", env=env)
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)


def test_init_dist_rejects_a_mismatched_world_size():
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    with pytest.raises(SystemExit, match="WORLD_SIZE"):
        bench.init_dist(2, "gloo")


def test_bench_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)
