"""pdm_gather_bev at the bench shape (bs=32, P=1024, C=128, 7x7, degree 2, 176x200): time, GB/s of the grid written."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pdm_ssd_amd import pdm_ops, synthetic
dev = torch.device("cuda:0")
g = pdm_ops.BevGrid(list(synthetic.KITTI_RANGE), (0.4, 0.4, 4.0))
for kind in ("uniform", "lidar"):
    gen = synthetic.uniform_clouds if kind == "uniform" else synthetic.lidar_like_clouds
    B, P, C = 32, 1024, 128
    xyz = torch.from_numpy(np.ascontiguousarray(gen(B, 16384, 3)[:, ::16, :3])).to(dev).contiguous()
    feat = torch.randn(B, P, C, device=dev); sh = torch.randn(B, P, 9, device=dev) * 0.1; inv = torch.rand(B, P, device=dev) + 0.5
    f = lambda: pdm_ops.pdm_gather(xyz, feat, sh, inv, g, (7, 7, 1), 2)
    a = f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{kind}: {ms*1e3:.1f} us  {a[0].numel()*4/1e9/(ms/1e3):.0f} GB/s of grid", flush=True)
