"""group_points forward/backward rate with RANDOM indices (worst case for the gather) and with ball-query indices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (C, N, M, ns) in [(96, 4096, 1024, 32), (256, 1024, 256, 32), (1, 16384, 4096, 32), (3, 16384, 4096, 32), (512, 256, 64, 32), (96, 4096, 1024, 16)]:
    f = torch.randn(32, C, N, device=dev, requires_grad=True)
    idx = torch.randint(0, N, (32, M, ns), dtype=torch.int32, device=dev)
    ms = t(lambda: pu.grouping_operation(f.detach(), idx))
    by = 32 * (4 * M * ns + 4 * C * N + 4 * C * M * ns)
    out = pu.grouping_operation(f, idx)
    go = torch.randn_like(out)
    msb = t(lambda: torch.autograd.grad(out, f, go, retain_graph=True))
    print(f"group_points C={C:4d} N={N:6d} M={M:5d} ns={ns}: fwd {ms*1e3:7.1f} us {by/1e9/(ms/1e3):6.0f} GB/s | bwd {msb*1e3:7.1f} us {by/1e9/(msb/1e3):6.0f} GB/s (random idx)")
