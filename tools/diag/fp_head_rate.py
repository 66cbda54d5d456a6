"""Last FP module + point head: one launch (pdm_fp_head_fused) against separate launches, at the bench shape with the
backbone's real three-NN indices.  Times the two modules' kernels alone (HIP events around the point head call with the FP
module deferred into it, or around FP module + head), settled clock."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native, fused
import bench
dev = torch.device("cuda:0")
model = bench.build_detector(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
_, points = bench.make_batch(B, 16384, kind, 1234, dev)
head, bb = model.point_head, model.backbone_3d
l = _native.lib()

def once(fusion, mask):
    head.use_fp_fusion = fusion
    old = l.pdm_tune_fp_chain_mask(mask)
    bd = {'batch_size': B, 'points': points, 'points_per_sample_checked': True, 'defer_last_fp': True}
    with torch.no_grad():
        bd = bb(bd)
        d = bd.get('point_features_deferred')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if not fusion:
            bd.pop('point_features_deferred').materialize()
        head(bd)
        e1.record()
        torch.cuda.synchronize()
    l.pdm_tune_fp_chain_mask(old)
    return e0.elapsed_time(e1) * 1e3

for _ in range(30):
    once(True, 2)
for name, fusion, mask in (("one launch", True, 2), ("FP1 chain kernel + pair launch", False, 3), ("FP1 LDS-tiled kernel + pair launch", False, 2)) * 2:
    ts = sorted(once(fusion, mask) for _ in range(15))
    print(f"{kind} bs={B}: last FP module + point head (+ decode), {name:36s}: median {ts[7]:8.1f} us  min {ts[0]:8.1f}", flush=True)
for n in (1, 2, 3, 4, 8, 16):
    old = l.pdm_tune_fp_head_tiles(n)
    ts = sorted(once(True, 2) for _ in range(15))
    l.pdm_tune_fp_head_tiles(old)
    print(f"{kind} bs={B}: one launch, {n:2d} tiles per workgroup: median {ts[7]:8.1f} us  min {ts[0]:8.1f}", flush=True)
