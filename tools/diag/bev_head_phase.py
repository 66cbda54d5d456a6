"""Phase clock (s_memtime) of the heat-map head's one-kernel form (rows_chain_kernel<8,4,4,1,true>): cycles a wave spends in
the depthwise prologue and in layers 1 / 2 / 3 of a tile, from the timing build `make -C pdm_ssd_amd/csrc diag-rc`
(librc_diag4.so writes timestamps of each workgroup's first two tiles instead of results).
MFMA floor per tile with two waves per SIMD: 2 x 32 cycles x (128 + 64 + 16) MFMAs = 13312 cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native
_native.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librc_diag4.so")
from pdm_ssd_amd.detector_config import build_pdm_ssd
dev = torch.device("cuda:0")
m = build_pdm_ssd().to(dev).eval()
B = 32
sf = torch.randn(B, 188, 188, 128, device=dev).permute(0, 3, 1, 2)
head = m.dense_head
with torch.no_grad():
    for _ in range(100):
        logits = head._fused_logits(sf)
    torch.cuda.synchronize()
raw = logits.permute(0, 2, 3, 1)          # (B, H, W, num_class) view of the (B, H, W, 4) buffer
buf = raw._base if raw._base is not None else raw
while buf._base is not None:
    buf = buf._base
cap = _native.lib().pdm_tune_rows_chain_dw_wg_per_cu(1)
_native.lib().pdm_tune_rows_chain_dw_wg_per_cu(cap)
nwg = 256 * cap     # the kernel's grid: 256 CUs x pdm_tune_rows_chain_dw_wg_per_cu (two resident per CU at a time)
ts = buf.contiguous().view(-1).view(torch.int64)[: nwg * 2 * 4].view(nwg, 2, 4).cpu()
d = (ts[:, :, 1:] - ts[:, :, :-1]).double()
for k in (0, 1):
    print(f"tile {k} of a workgroup: layer 1 / 2 / 3 mean cycles", [round(float(d[:, k, i].mean())) for i in range(3)],
          " min", [int(d[:, k, i].min()) for i in range(3)])
gap = (ts[:, 1, 0] - ts[:, 0, 3]).double()
print("end of tile 0 -> layer 1 of tile 1 (depthwise prologue of tile 1):", round(float(gap.mean())), "cycles, min", int(gap.min()))
print("tile 1 whole (prologue + layers):", round(float((ts[:, 1, 3] - ts[:, 0, 3]).double().mean())))
dw = buf.contiguous().view(-1).view(torch.int64)[nwg * 8: nwg * 8 + nwg * 9].view(nwg, 9).cpu()
dd = (dw[:, 1:] - dw[:, :-1]).double()
names = ["halo write (waits for the halo loads)", "barrier 1", "taps + crossing write", "barrier 2", "crossing read + halo write 2", "barrier 1", "taps", "barrier 2"]
print("depthwise prologue of tile 1, wave 0 (two stages of 4 slices):")
for i, nm in enumerate(names):
    print(f"  {nm:42s} {float(dd[:, i].mean()):8.0f}  (min {int(dd[:, i].min())})")
