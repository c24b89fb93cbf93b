"""The N>1 path of bench.py on CPU: world_size 2 over gloo.

The hot path does not shard (DESIGN.md "Multi-GPU": replicas only), so the only distributed code is the
timing contract -- barrier on both sides of the timed region, MAX over ranks, whole-job value = N x per-rank
work.  This test runs exactly that code (bench.timed_region / bench.whole_job_value) with two processes."""
import os
import socket
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}

    def step():  # rank 1 is the slow replica: the job time must be ITS time
        calls["n"] += 1
        time.sleep(0.02 if rank == 0 else 0.06)

    elapsed = bench.timed_region(step, steps=5, warmup=2, world=world, sync=lambda: None, device="cpu")
    q.put((rank, elapsed, calls["n"], bench.whole_job_value(1e12, 5, world, elapsed)))
    dist.barrier()
    dist.destroy_process_group()


def test_timed_region_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, c0, v0), (r1, e1, c1, v1) = res
    assert (r0, r1) == (0, 1)
    assert c0 == c1 == 7                      # 2 warm-up + exactly 5 timed steps on every rank
    assert e0 == pytest.approx(e1, abs=1e-12)  # MAX over ranks: identical on all ranks
    assert 0.29 < e0 < 0.6                    # the slow rank's 5 x 60 ms, not the fast rank's 5 x 20 ms
    assert v0 == pytest.approx(2 * 1e12 * 5 / e0 / 1e12)   # whole-job aggregate: 2 replicas


def test_single_process_path():
    import bench
    n = {"c": 0}

    def step():
        n["c"] += 1
    e = bench.timed_region(step, steps=3, warmup=1, world=1, sync=lambda: None, device="cpu")
    assert n["c"] == 4 and e >= 0
    assert bench.whole_job_value(2.0e12, 3, 1, 1.5) == pytest.approx(4.0)
