import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pdm_ssd_amd import synthetic
from pdm_ssd_amd.pointnet2_batch import pointnet2_modules as pm
dev = torch.device("cuda:0")
torch.manual_seed(2)
for mlps, ns in (([[1, 16, 16, 32]], [16]), ([[1, 32, 32, 64]], [32]), ([[1, 16, 16, 32], [1, 32, 32, 64]], [16, 32])):
    sa = pm.PointnetSAModuleMSG(npoint=200, radii=[0.9, 1.8][:len(ns)], nsamples=ns, mlps=copy.deepcopy(mlps)).eval().to(dev)
    cl = synthetic.lidar_like_clouds(2, 1500, 11)
    xyz = torch.from_numpy(np.ascontiguousarray(cl[:, :, :3])).to(dev)
    feat = torch.randn(2, 1, 1500, device=dev)
    with torch.no_grad():
        _, a = sa(xyz, feat)
        sa.use_fused = False
        _, b = sa(xyz, feat)
    d = (a - b).abs()
    bad = d > 1e-4
    print(mlps, ns, "bad", int(bad.sum()), "of", bad.numel(), "per-channel bad:", bad.sum(dim=(0, 2)).tolist())
