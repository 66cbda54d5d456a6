"""pdm_gather_bev on the bench workload (bs = 32 lidar-like clouds, the 1024 points of SA level 2 per cloud, C = 128, 7x7 window on
the 176 x 200 map): time per call, the grid's write rate, points per tile.  `--lib FILE` times a PG_DIAG build
(`make -C pdm_ssd_amd/csrc diag-pg`: 1 = no accumulation, 2 = no weight arithmetic, 4 = stores only; results wrong by construction)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native, pdm_ops
if len(sys.argv) > 2 and sys.argv[1] == "--lib":
    _native.LIB_PATH = os.path.abspath(sys.argv[2])
    print("library:", _native.LIB_PATH, flush=True)
import bench
dev = torch.device("cuda:0")
backbone, neck = bench.build_models(dev)
_, pts = bench.make_batch(32, 16384, "lidar", 0, dev)
with torch.no_grad():
    bd = backbone({'batch_size': 32, 'points': pts})
    xyz = bd['sa_xyz'][neck.source_layer].contiguous()
B, P = xyz.shape[:2]
torch.manual_seed(0)
C, nsh = neck.feature_dim, neck.nsh
feat = torch.randn(B, P, C, device=dev)
sh = torch.randn(B, P, nsh, device=dev) * 0.2
sh[..., 0] += 3.5
inv2s2 = torch.rand(B, P, device=dev) * 0.5 + 0.2
g = neck.grid
print(f"B={B} P={P} C={C} grid {g.H}x{g.W}x{g.D} window {neck.dilation} degree {neck.degree}", flush=True)
# points per 8x8-cell tile (the lists pdm_bin_kernel builds): a point joins every tile its 7x7 window overlaps
bxy = torch.floor((xyz[..., :2] - torch.tensor(g.origin[:2], device=dev)) * torch.tensor(g.inv_cell[:2], device=dev)).long()
TW, TH = (g.W + 7) // 8, (g.H + 7) // 8
cnt = torch.zeros(B, TH, TW, dtype=torch.long, device=dev)
hx, hy = neck.dilation[0] // 2, neck.dilation[1] // 2
for ty in range(TH):
    oky = (bxy[..., 1] + hy >= ty * 8) & (bxy[..., 1] - hy <= ty * 8 + 7)
    for tx in range(TW):
        cnt[:, ty, tx] = (oky & (bxy[..., 0] + hx >= tx * 8) & (bxy[..., 0] - hx <= tx * 8 + 7)).sum(1)
c = cnt.flatten().float()
print(f"tiles {c.numel()}: empty {float((c == 0).float().mean()):.3f}, mean {float(c.mean()):.2f}, median {float(c.median()):.0f}, "
      f"p90 {float(c.quantile(0.9)):.0f}, p99 {float(c.quantile(0.99)):.0f}, max {int(c.max())}; > 32: {float((c > 32).float().mean()):.4f}", flush=True)
ref, wref = pdm_ops.pdm_gather(xyz, feat, sh, inv2s2, g, neck.dilation, neck.degree)
for _ in range(300): pdm_ops.pdm_gather(xyz, feat, sh, inv2s2, g, neck.dilation, neck.degree)    # the clock settles under load
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): out, w = pdm_ops.pdm_gather(xyz, feat, sh, inv2s2, g, neck.dilation, neck.degree)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(f"pdm_gather_bev (bin + gather): {ms * 1e3:.1f} us per call, grid {out.numel() * 4 / 1e6:.0f} MB -> {out.numel() * 4 / ms / 1e9:.2f} TB/s; "
      f"cells with weight: {float((w != 0).float().mean()):.3f}; reproducible: {torch.equal(out, ref)}", flush=True)
if len(sys.argv) <= 2:
    # against the scatter form (atomics, arrival order): the same sum up to summation order
    sgrid, swsum = pdm_ops.pdm_scatter(xyz, feat, sh, inv2s2, g, neck.dilation, neck.degree, 1) if hasattr(pdm_ops, "pdm_scatter") else (None, None)
    if sgrid is not None:
        sn = pdm_ops.bev_normalize(sgrid, swsum, C, g)
        print(f"max |gather - scatter form| = {float((out - sn.view_as(out)).abs().max()):.3e}", flush=True)
