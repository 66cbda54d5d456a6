"""Per-call time of the 8 ball_query + 16 group_points launches of the API-exact section of bench.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev = torch.device("cuda:0")
backbone, neck = bench.build_models(dev)
B = 32
_, points = bench.make_batch(B, 16384, "uniform", 1234, dev)
xyz = points[:, 1:4].contiguous().view(B, -1, 3)
feats = points[:, 4:].contiguous().view(B, -1, 1).permute(0, 2, 1).contiguous()
chans = [1, 96, 256, 512]
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0.0; totb = 0
with torch.no_grad():
    for k, m in enumerate(backbone.SA_modules):
        new_xyz = m.sample(xyz)
        f = feats if k == 0 else torch.randn(B, chans[k], xyz.shape[1], device=dev)
        xt = xyz.transpose(1, 2).contiguous()
        N, M = xyz.shape[1], new_xyz.shape[1]
        for g in m.groupers:
            ns = g.nsample
            us = t(lambda: pu.ball_query(g.radius, ns, xyz, new_xyz)); by = B * (12 * N + 12 * M + 4 * M * ns)
            idx = pu.ball_query(g.radius, ns, xyz, new_xyz)
            print(f"SA{k+1} r={g.radius:<4} ball_query N={N:6d} M={M:5d} ns={ns}: {us:7.1f} us {by/1e3/us:7.0f} GB/s"); tot += us; totb += by
            for name, src in (("xyz ", xt), ("feat", f)):
                C = src.shape[1]
                us = t(lambda: pu.grouping_operation(src, idx)); by = B * (4 * M * ns + 4 * C * N + 4 * C * M * ns)
                print(f"      group {name} C={C:4d}                          : {us:7.1f} us {by/1e3/us:7.0f} GB/s"); tot += us; totb += by
        xyz = new_xyz
print(f"total {tot:.1f} us, {totb/1e6:.1f} MB -> {totb/1e3/tot:.0f} GB/s = {totb/1e3/tot/8000:.3f} of 8 TB/s")
