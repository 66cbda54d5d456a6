"""Throughput of the GPU input path: 32 raw KITTI-sized clouds (~120k points) -> (32*16384, 5) points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pdm_ssd_amd import input_path as ip
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
clouds = []
for b in range(32):
    n = int(rng.integers(110000, 125000))
    r = np.where(rng.uniform(size=n) < 0.15, rng.uniform(40, 70, n), rng.uniform(1, 39.9, n)); th = rng.uniform(-np.pi, np.pi, n)
    clouds.append(np.stack([r * np.cos(th), r * np.sin(th), rng.uniform(-3, 1, n), rng.uniform(0, 1, n)], 1).astype(np.float32))
raw, cnt, host = ip.upload_raw(clouds, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    raw, cnt, host = ip.upload_raw(clouds, dev)
torch.cuda.synchronize()
up = (time.perf_counter() - t0) / 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ip.sample_points_batch(raw, cnt, 16384, seed=1); torch.cuda.synchronize()
e0.record()
for _ in range(20):
    ip.sample_points_batch(raw, cnt, 16384, seed=1)
e1.record(); torch.cuda.synchronize()
ks = e0.elapsed_time(e1) / 20
print(f"raw rows {raw.shape[0]}  upload (pinned staging + H2D, host wall) {up*1e3:.2f} ms = {raw.numel()*4/up/1e9:.1f} GB/s; "
      f"sampler kernel {ks:.3f} ms per batch of 32 = {32/ks*1e3:.0f} clouds/s")
# the numpy reference rule on one core, for scale
def ref_rule(points, num_points=16384):
    d = np.linalg.norm(points[:, 0:3], axis=1); near = d < 40.0
    far_idx = np.where(near == 0)[0]; near_idx = np.where(near == 1)[0]
    ch = np.concatenate((np.random.choice(near_idx, num_points - len(far_idx), replace=False), far_idx)) \
        if num_points > len(far_idx) else np.random.choice(np.arange(len(points)), num_points, replace=False)
    np.random.shuffle(ch)
    return points[ch]
t0 = time.perf_counter()
for c in clouds[:8]:
    ref_rule(c)
print(f"numpy rule (data_processor.py:189-210), one core: {(time.perf_counter()-t0)/8*1e3:.2f} ms per cloud")
