"""three_interpolate backward on the four FP shapes of the training step (bs=32): time, GB/s of grad_out read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as hipops
dev = torch.device("cuda:0")
B = 32
for c, n, m in [(256, 16384, 4096), (512, 4096, 1024), (512, 1024, 256), (1024, 256, 64)]:
    grad_out = torch.randn(B, c, n, device=dev)
    idx = torch.randint(0, m, (B, n, 3), dtype=torch.int32, device=dev)
    w = torch.rand(B, n, 3, device=dev)
    gp = torch.zeros(B, c, m, device=dev)
    def run():
        hipops.three_interpolate_grad_wrapper(B, c, n, m, grad_out, idx, w, gp)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"c={c:5d} n={n:6d} m={m:5d}: {ms*1e3:8.1f} us   {grad_out.numel()*4/ms/1e6:7.1f} GB/s of grad_out")
