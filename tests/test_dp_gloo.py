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
