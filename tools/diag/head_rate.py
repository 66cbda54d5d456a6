"""Hybrid head at the bench shape (bs=32): per-part time, eval mode, fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd.detector_config import build_pdm_ssd
dev = torch.device("cuda:0")
m = build_pdm_ssd().to(dev).eval()
B, N = 32, 16384
pf = torch.randn(B * N, 128, device=dev)
pc = torch.cat([torch.arange(B, device=dev).repeat_interleave(N)[:, None].float(), torch.rand(B * N, 3, device=dev) * 40], 1)
sf = torch.randn(B, 200, 176, 128, device=dev).permute(0, 3, 1, 2)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    print("point head (fused rows kernels) ms", t(lambda: m.point_head({'batch_size': B, 'point_features': pf, 'point_coords': pc})))
    m.point_head.use_fused = False
    print("point head (torch layers)       ms", t(lambda: m.point_head({'batch_size': B, 'point_features': pf, 'point_coords': pc})))
    print("heat-map head                   ms", t(lambda: m.dense_head({'spatial_features': sf})))
    x = sf.contiguous(memory_format=torch.channels_last)
    for i, layer in enumerate(m.dense_head.shared_conv):
        y = layer(x); ms = t(lambda: layer(x)); print("   shared_conv", i, type(layer).__name__, round(ms, 3)); x = y
