"""The N > 1 path of bench.py on CPU: world_size 2 over gloo, rendezvous on 127.0.0.1.  The data path shards by batch
and exchanges nothing; the only collective is the all_gather of {elapsed, FLOPs} (bench.gather_counters)."""
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


def test_gpus_flag_requires_matching_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)
