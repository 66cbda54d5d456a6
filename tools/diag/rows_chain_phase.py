"""Phase clock (s_memtime) of the point head's chain kernel: cycles a wave spends in layers 1 / 2 / 3 of a tile, from
the timing build `make -C pdm_ssd_amd/csrc diag-rc` (librc_diag4.so writes timestamps instead of results).
With two waves per SIMD sharing the MFMA pipe the floor per layer is 2 x 32 cycles x MFMAs: 32768 / 65536 / 4096."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native, fused
_native.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librc_diag4.so")
from pdm_ssd_amd.dense_heads.point_head_box import _fc_layers
from pdm_ssd_amd.dense_heads.point_head_template import PointHeadTemplate
dev = torch.device("cuda:0")
torch.manual_seed(0)
rows = 32 * 16384
seq = PointHeadTemplate.make_fc_layers([256, 256], 128, 3).to(dev).eval()
pk = fused.PackedMLP(_fc_layers(seq), dev)
x = torch.randn(rows, 128, device=dev)
out = torch.zeros(rows, 4, device=dev)
for _ in range(200):   # settled clock
    fused.rows_forward(pk, x, out, relu_last=False)
torch.cuda.synchronize()
ts = out.view(-1).view(torch.int64)[: 3072 * 2 * 4].view(3072, 2, 4).cpu()
d = (ts[:, :, 1:] - ts[:, :, :-1]).double()
for k in (0, 1):
    print(f"tile {k} of a workgroup: layer 1 / 2 / 3 mean cycles", [round(float(d[:, k, i].mean())) for i in range(3)],
          " min", [int(d[:, k, i].min()) for i in range(3)], " max", [int(d[:, k, i].max()) for i in range(3)])
gap = (ts[:, 1, 0] - ts[:, 0, 3]).double()
print("end of tile 0 -> layer 1 of tile 1 (stores, next rows):", round(float(gap.mean())), "cycles")
print("whole tile 0 (layers only):", round(float((ts[:, 0, 3] - ts[:, 0, 0]).double().mean())), " floor 102400")
t0 = ts[:, 0, 0].min()
print("by launch order (512 workgroups are resident at a time): start of tile 0 [k cycles after the first], layer-2 cycles of tile 0 / tile 1")
for g0 in range(0, 3072, 256):
    sl = slice(g0, g0 + 256)
    print(f"  workgroups {g0:4d}..{g0 + 255:4d}: start {float((ts[sl, 0, 0] - t0).double().mean()) / 1e3:8.1f}   "
          f"{float(d[sl, 0, 1].mean()):8.0f} / {float(d[sl, 1, 1].mean()):8.0f}   layer 1: {float(d[sl, 0, 0].mean()):8.0f} / {float(d[sl, 1, 0].mean()):8.0f}")

