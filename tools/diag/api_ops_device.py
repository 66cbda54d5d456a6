"""DEVICE time of each of the 8 ball_query (+ grid builds) + 16 group_points launches of bench.py's API-exact section:
every call captured 10x into a hipGraph and replayed (no Python between the launches), so a launch's figure is its
kernel time plus one dependent-launch boundary.  Also the whole 26-call sequence as one graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
backbone, neck = bench.build_models(dev)
B = 32
_, points = bench.make_batch(B, 16384, kind, 1234, dev)
xyz = points[:, 1:4].contiguous().view(B, -1, 3)
feats = points[:, 4:].contiguous().view(B, -1, 1).permute(0, 2, 1).contiguous()
chans = [1, 96, 256, 512]

def graph_us(fn, reps=10, replays=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * replays) * 1e3

plan = []
tot = 0.0; totb = 0
with torch.no_grad():
    for k, m in enumerate(backbone.SA_modules):
        new_xyz = m.sample(xyz)
        f = feats if k == 0 else torch.randn(B, chans[k], xyz.shape[1], device=dev)
        xt = xyz.transpose(1, 2).contiguous()
        N, M = xyz.shape[1], new_xyz.shape[1]
        for g in m.groupers:
            ns = g.nsample
            plan.append((g.radius, ns, xyz, new_xyz, f, xt))
            us = graph_us(lambda: pu.ball_query(g.radius, ns, xyz, new_xyz)); by = B * (12 * N + 12 * M + 4 * M * ns)
            idx = pu.ball_query(g.radius, ns, xyz, new_xyz)
            print(f"SA{k+1} r={g.radius:<4} ball_query (own grid build where N >= 2048) N={N:6d} M={M:5d} ns={ns}: {us:7.1f} us"); tot += us; totb += by
            for name, src in (("xyz ", xt), ("feat", f)):
                C = src.shape[1]
                us = graph_us(lambda: pu.grouping_operation(src, idx)); by = B * (4 * M * ns + 4 * C * N + 4 * C * M * ns)
                print(f"      group {name} C={C:4d}: {us:7.1f} us {by/1e3/us:7.0f} GB/s"); tot += us; totb += by
        xyz = new_xyz
    print(f"sum of single-call graphs {tot:.1f} us (every ball_query with its own grid build), {totb/1e6:.1f} MB")
    def whole():
        with pu.shared_search_grids():
            for radius, ns, x, nx, f, xt in plan:
                idx = pu.ball_query(radius, ns, x, nx)
                pu.grouping_operation(xt, idx)
                pu.grouping_operation(f, idx)
    us = graph_us(whole, reps=4, replays=10)
    print(f"whole sequence (8 ball_query + 2 shared grid builds + 16 group_points) as one graph: {us:.1f} us -> {totb/1e3/us:.0f} GB/s = {totb/1e3/us/8000:.3f} of 8 TB/s")
