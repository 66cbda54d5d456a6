"""Data-parallel plumbing: one process per GPU, whole clouds sharded across ranks.

The reference trains/evaluates with torch.distributed (backend 'nccl', hard-coded at
/root/reference/tools/train.py:75 and tools/test.py:149; init at pcdet/utils/common_utils.py:189-204) and
shards whole samples with a DistributedSampler (pcdet/datasets/__init__.py:69-74).  On ROCm the same
backend string resolves to RCCL over xGMI.  The hot path itself has NO data-path collective: every kernel's
outermost dimension is the sample index.  Only training adds one exchange per step (gradient all-reduce,
handled by DistributedDataParallel) and the benchmark adds its timing barrier / max-reduce.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world, local_rank).  Initialises the process group when WORLD_SIZE > 1; rendezvous on
    127.0.0.1 unless MASTER_ADDR is set.  backend defaults to 'nccl' (= RCCL) with a GPU, 'gloo' without."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local_rank


def shard_range(total, rank, world):
    """Contiguous shard [begin, end) of `total` whole clouds for `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(value, device=None):
    """MAX of a python float over all ranks (the benchmark's elapsed time)."""
    if not dist.is_initialized():
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def average_gradients(module):
    """Flat-bucket gradient all-reduce (SUM then /world): what DDP does for this model in one ~12 MB bucket;
    kept as an explicit function so the exchange step can be tested on CPU with gloo."""
    if not dist.is_initialized():
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
