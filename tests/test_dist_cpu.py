"""CPU, world_size 2 over gloo: the multi-process plumbing of bench.py (barrier + max-over-ranks timing,
batch-sharded replicas with no data-path collective)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    r, w, local, d = bench.dist_setup(world)       # gloo here (no GPU); nccl (= RCCL) on the GPU box
    assert (r, w) == (rank, world) and d is not None and d.get_backend() == "gloo"
    bench.barrier(d)
    local_seconds = 1.0 + rank                      # rank 1 is the straggler
    t = bench.max_over_ranks(local_seconds, d, torch.device("cpu"))
    # whole-job value: every rank processed B units per step on its own shard; time = slowest rank
    B, steps = 32, 10
    value = w * B * steps / t
    q.put((rank, t, value))
    d.destroy_process_group()


def test_two_rank_timing_and_aggregate():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [2.0, 2.0]                      # MAX over ranks on both
    assert res[0][2] == pytest.approx(2 * 32 * 10 / 2.0)            # aggregate tokens/s over the whole job


def test_bytes_model_matches_survey():
    sys.path.insert(0, ROOT)
    import bench
    # C2 = 209 977 344 algorithmic bytes per qK^T launch, same for sV at Tq = 4096
    assert bench.kgemv_bytes(32, 32, 32, 128, 4096, 32, 2) == 209_977_344
    assert bench.vgemv_bytes(32, 32, 32, 128, 4096, 32, 2) == 209_977_344


def test_gpus_flag_requires_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit):
        bench.dist_setup(2)
