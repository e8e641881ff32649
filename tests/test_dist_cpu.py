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


def test_world_size_mismatch_is_refused(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("WORLD_SIZE", "3")
    with pytest.raises(SystemExit):
        bench.dist_setup(2)


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_main_starts_its_own_ranks(launcher):
    """`python bench.py --gpus 2` (the form the driver uses) must work without a wrapper: main() re-executes itself once
    per rank, the ranks rendezvous over 127.0.0.1 (gloo here, RCCL on GPUs), rank 0 prints ONE JSON line with the
    whole-job value, the job's time = the slowest rank.  --dry-run swaps the GPU work for a sleep.  The same file under
    torch.distributed.run must give the same shape of answer."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--dry-run"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + os.getpid() % 200)] + cmd[1:]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "weak" and len(out["per_rank_ms_per_step"]) == 2
    assert out["ms_per_step"] == pytest.approx(max(out["per_rank_ms_per_step"]), rel=0.2)
    assert out["per_rank_ms_per_step"][1] > out["per_rank_ms_per_step"][0] * 0.9      # rank 1 sleeps twice as long per step
    assert out["value"] == pytest.approx(2 * 32 * 6 / (out["ms_per_step"] * 6e-3), rel=1e-3)


def test_a_failing_rank_fails_the_launch():
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--bits", "3"],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
