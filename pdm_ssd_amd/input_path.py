"""GPU-side input path: `.bin` reader, a batched `sample_points` with the reference's
near/far rule, the batch-index column of `collate_batch`, and a pinned-memory upload — the pieces between a KITTI
velodyne file and the `points (B*N, 1+C)` tensor PointNet2MSG consumes
(kitti_dataset.py:63-66, data_processor.py:182-212, dataset.py:237-244, pcdet/models/__init__.py:24-38).

The reference samples per cloud on CPU workers with numpy's global RNG; at thousands of frames per second per GPU that
is the bottleneck, and its draws cannot be replayed elsewhere.  Here the whole batch is sampled by one HIP kernel and
the draw is a deterministic function of (seed, cloud index, raw row) — see include/pdmssd_hip.h / DESIGN.md section 10.
There is no CPU fallback.
"""
import numpy as np
import torch

from . import _native

MAX_NUM_POINTS = 16384


def read_velodyne_bin(path, num_features=4):
    """(N, num_features) float32 rows [x, y, z, intensity] of a KITTI velodyne file (kitti_dataset.py:63-66)."""
    return np.fromfile(str(path), dtype=np.float32).reshape(-1, num_features)


def upload_raw(clouds, device):
    """list of (N_i, C) float32 arrays -> (raw (sum N_i, C) device tensor, counts (B,) int32 device tensor,
    host counts).  One pinned staging buffer, one asynchronous copy (load_data_to_gpu's role)."""
    counts = [int(c.shape[0]) for c in clouds]
    C = int(clouds[0].shape[1])
    stage = torch.empty((sum(counts), C), dtype=torch.float32, pin_memory=True)
    o = 0
    for c, n in zip(clouds, counts):
        stage[o:o + n] = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32))
        o += n
    raw = stage.to(device, non_blocking=True)
    cnt = torch.tensor(counts, dtype=torch.int32).to(device, non_blocking=True)
    return raw, cnt, counts


def sample_points_batch(raw, counts, num_points, seed=0, host_counts=None, return_choice=False):
    """raw (sum N_i, C) rows [x, y, z, ...] on the GPU, counts (B,) int32 on the GPU -> points (B * num_points, 1 + C)
    rows [cloud, x, y, z, ...] ready for PointNet2MSG.forward.

    Per cloud (data_processor.py:189-210): with more than num_points raw points every point at depth >= 40 m is kept
    and the rest of the quota is drawn from the near points (or, if the far points alone exceed the quota, num_points
    are drawn from all points); shorter clouds are kept whole and padded with distinct extra picks; the result is
    shuffled.  `host_counts` (optional list) lets the call verify what the reference would reject."""
    assert raw.is_cuda and raw.dtype == torch.float32 and raw.is_contiguous() and raw.dim() == 2
    assert counts.is_cuda and counts.dtype == torch.int32 and counts.is_contiguous() and counts.dim() == 1
    if not 1 <= num_points <= MAX_NUM_POINTS:
        raise ValueError(f"num_points={num_points}: this build samples at most {MAX_NUM_POINTS} points per cloud")
    B, C = counts.numel(), raw.shape[1]
    if host_counts is not None:
        if sum(host_counts) != raw.shape[0]:
            raise ValueError("counts do not sum to the number of raw rows")
        for n in host_counts:
            if n < 1 or num_points - n > n:
                raise ValueError(f"a cloud of {n} points cannot be padded to {num_points} without replacement")
    out = torch.empty((B * num_points, 1 + C), dtype=torch.float32, device=raw.device)
    choice = torch.empty((B * num_points,), dtype=torch.int32, device=raw.device) if return_choice else None
    _native.call("pdm_sample_points", torch.cuda.current_stream(raw.device).cuda_stream, B, int(num_points),
                 int(seed) & 0xffffffff, C, raw.data_ptr(), counts.data_ptr(), out.data_ptr(),
                 0 if choice is None else choice.data_ptr())
    return (out, choice) if return_choice else out
