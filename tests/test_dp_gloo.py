"""Multi-process data-parallel path on CPU (gloo, world_size 2): rendezvous from the environment on
127.0.0.1, whole-cloud sharding, the benchmark's barrier + MAX-over-ranks timing, and the gradient
all-reduce exchange step of training."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist

    from pdm_ssd_amd import dist_utils, synthetic
    r, w, _ = dist_utils.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    # whole clouds sharded: global batch of 6 clouds, rank r generates only its own (seed = 1234 + global index)
    b0, b1 = dist_utils.shard_range(6, r, w)
    mine = synthetic.uniform_clouds(b1 - b0, 64, seed0=1234 + b0)
    checksum = torch.tensor([float(mine.sum())], dtype=torch.float64)
    dist.all_reduce(checksum)  # test-only collective: union of the shards == the unsharded batch
    dist_utils.barrier()
    t = dist_utils.max_over_ranks(1.0 + rank)  # slowest rank defines the step time
    # gradient exchange: every rank ends with the mean gradient
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 3)
    x = torch.full((2, 4), float(rank + 1))
    lin(x).sum().backward()
    local = lin.weight.grad.clone()
    dist_utils.average_gradients(lin)
    q.put((rank, b0, b1, float(checksum.item()), t, local.numpy(), lin.weight.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel():
    from pdm_ssd_amd import synthetic
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 6)]
    full = float(synthetic.uniform_clouds(6, 64).sum())
    assert all(abs(r[3] - full) < 1e-3 * abs(full) for r in res)
    assert all(r[4] == 2.0 for r in res)  # MAX over ranks
    mean_grad = (res[0][5] + res[1][5]) / 2
    for r in res:
        np.testing.assert_allclose(r[6], mean_grad, rtol=1e-6)


def _run_bench(*argv, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_self_launches_ranks_gloo():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts two ranks itself (one process per GPU, as
    tools/train.py:74-76 / common_utils.py:189-204 expect them), they rendezvous on 127.0.0.1, run the barrier +
    MAX-over-ranks protocol and rank 0 prints exactly one JSON line."""
    import json
    r = _run_bench("--gpus", "2", "--steps", "3", "--rendezvous-only")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["backend"] == "gloo"
    # rank r sleeps (r + 1) ms per step: the MAX over ranks is the slower rank's time
    assert line["ms_per_step"] >= 2.0


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour of the launcher")
def test_bench_launcher_relays_failure_without_gpu():
    """Without a GPU every child stops at the 'no GPU' check (the hot path has no CPU fallback); the launcher must
    come back with a non-zero code instead of an AssertionError or a hang."""
    r = _run_bench("--gpus", "2", "--steps", "1")
    assert r.returncode != 0
    assert "no GPU visible" in r.stderr and "Traceback" not in r.stderr
    r1 = _run_bench("--steps", "1")
    assert r1.returncode != 0 and "no GPU visible" in r1.stderr
