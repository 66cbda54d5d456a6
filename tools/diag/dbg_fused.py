import sys, copy, numpy as np, torch
sys.path.insert(0, '/root/repo')
from pdm_ssd_amd import synthetic, _native
from pdm_ssd_amd.pointnet2_batch import pointnet2_modules as pm
dev = torch.device('cuda:0')
torch.manual_seed(2)
for forced in (1, 2):
    _native.lib().pdm_tune_fused_tiles(forced)
    sa = pm.PointnetSAModuleMSG(npoint=200, radii=[0.9, 1.8], nsamples=[16, 32], mlps=[[1, 16, 16, 32], [1, 32, 32, 64]]).to(dev).eval()
    cl = synthetic.lidar_like_clouds(2, 1500, 11)
    xyz = torch.from_numpy(np.ascontiguousarray(cl[:, :, :3])).to(dev)
    feat = torch.randn(2, 1, 1500, device=dev)
    with torch.no_grad():
        nx, nf = sa(xyz, feat)
        sa.use_fused = False
        _, nf2 = sa(xyz, feat)
    d = (nf - nf2).abs()
    bad = (d > 1e-4)
    print("NT", forced, "bad frac", bad.float().mean().item(), "per-channel bad", bad.any(0).any(-1).nonzero().flatten().tolist()[:40])
    print("   bad centres (b=0)", bad[0].any(0).nonzero().flatten().tolist()[:30])
