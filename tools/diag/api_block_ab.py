"""A/B of the grid ball-query forms on BASELINE's target block (see api_block.py), all in ONE process on one box:
form 1 (quad), 2 (lane; centres per wave 16 / 64), 3 (per cloud by density) x uniform / lidar-like clouds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native
if "--lib" in sys.argv:
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    print("library:", _native.LIB_PATH, flush=True)
import bench
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev = torch.device("cuda:0")
l = _native.lib()
backbone, neck = bench.build_models(dev)
B = 32
chans = [1, 96, 256, 512]
def block(kind):
    _, points = bench.make_batch(B, 16384, kind, 1234, dev)
    xyz = points[:, 1:4].contiguous().view(B, -1, 3)
    feats = points[:, 4:].contiguous().view(B, -1, 1).permute(0, 2, 1).contiguous()
    plan = []
    for k, m in enumerate(backbone.SA_modules):
        new_xyz = m.sample(xyz)
        f = feats if k == 0 else torch.randn(B, chans[k], xyz.shape[1], device=dev)
        xt = xyz.transpose(1, 2).contiguous()
        for g in m.groupers:
            plan.append((g.radius, g.nsample, xyz, new_xyz, f, xt))
        xyz = new_xyz
    def whole():
        with pu.shared_search_grids():
            for radius, ns, x, nx, f, xt in plan:
                idx = pu.ball_query(radius, ns, x, nx)
                pu.grouping_operation(xt, idx)
                pu.grouping_operation(f, idx)
    return whole
def graph_us(fn, replays=40):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / replays * 1e3
with torch.no_grad():
    for kind in ("uniform", "lidar"):
        fn = block(kind)
        for rep in range(2):
            forms = (("quad", 1, 16, 96, 16), ("auto", 3, 16, 96, 16)) if "--lib" in sys.argv else \
                (("quad", 1, 16, 96, 16), ("quad, 4-wave small scans", 1, 16, 96, 4), ("lane cpw16", 2, 16, 96, 16), ("lane cpw16 heavy32", 2, 16, 32, 16),
                 ("lane cpw64", 2, 64, 96, 16), ("by density", 3, 16, 96, 16))
            if "--group-nt" in sys.argv:
                forms = (("quad", 1, 16, 96, 16), ("quad, non-temporal group stores", 1, 16, 96, 16 + 256))
            for name, form, cpw, heavy, sw in forms:
                l.pdm_tune_group_nt(1 if sw >= 256 else 0); sw &= 255
                l.pdm_tune_bq_quad(form); l.pdm_tune_bq_cpw(cpw); l.pdm_tune_bq_heavy(heavy); l.pdm_tune_bq_small_waves(sw)
                us = graph_us(fn)
                print(f"{kind:8s} {name:34s}: {us:7.1f} us = {1676.79e3 / us / 8000:.3f} of 8 TB/s", flush=True)
