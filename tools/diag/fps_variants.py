"""FPS kernel shapes side by side (pdm_tune_fps_variant): time per call and per iteration at bs=32, and that every
variant returns the indices of round 1's kernel (variant 4 / 16), which the parity tests hold to the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pdm_ssd_amd import _native, synthetic
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev = torch.device('cuda:0'); l = _native.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CASES = [(16384, 4096, (4, 5, 7, 9, 10, 0)), (8192, 2048, (16, 17, 18, 19, 0)), (4096, 1024, (16, 17, 18, 19)),
         (2048, 512, (16, 17, 18, 19)), (1024, 256, (16, 17, 18))]
for kind in ('uniform', 'lidar'):
    gen = synthetic.uniform_clouds if kind == 'uniform' else synthetic.lidar_like_clouds
    for N, m, variants in CASES:
        xyz = torch.from_numpy(np.ascontiguousarray(gen(B, N, 5)[:, :, :3])).to(dev)
        ref = None
        for v in variants:
            l.pdm_tune_fps_variant(v)
            idx = pu.furthest_point_sample(xyz, m); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                idx = pu.furthest_point_sample(xyz, m)
            e1.record(); torch.cuda.synchronize()
            if ref is None:
                ref = idx.clone()
            ms = e0.elapsed_time(e1) / 3
            print(f"{kind:8s} N={N:6d} m={m:5d} variant {v:2d}: {ms:7.3f} ms  {ms * 1e3 / (m - 1):.3f} us/iter  same={torch.equal(idx, ref)}", flush=True)
l.pdm_tune_fps_variant(0)
