"""Score-ranked sampling (pdm_topk_sampling) against FPS on the SA sampling shapes of the bench (bs=32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N, m in [(16384, 4096), (4096, 1024), (1024, 256), (256, 64)]:
    xyz = torch.rand(32, N, 3, device=dev) * 50
    score = torch.rand(32, N, device=dev)
    t_topk = timed(lambda: pu.topk_sample(score, m))
    t_torch = timed(lambda: torch.topk(score, m, dim=-1))
    t_fps = timed(lambda: pu.furthest_point_sample(xyz, m), reps=3)
    print(f"N={N:6d} -> {m:5d}: pdm_topk_sampling {t_topk:8.1f} us   torch.topk {t_torch:8.1f} us   FPS {t_fps:9.1f} us")
