"""CPU statement of the module-level hot path: PointNet2MSG (4 SA + 4 FP) + PDM neck.

TEST INFRASTRUCTURE ONLY (same rule as cpu_oracle.py): used by tests/ as the checker of the GPU
modules and by bench.py's cpu_baseline leg as the timed "CPU fallback" stand-in (the reference has
no CPU path for these operators).

Operators = oracle/pointnet2_oracle.c / pdm_oracle.c through cpu_oracle.py; the dense contractions
(Conv2d 1x1 + BatchNorm + ReLU, max-pool, inverse-distance weights) run in torch on the CPU with the
weights of the module under test, following
/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:19-55 (SA) and :141-170 (FP)
and /root/reference/pcdet/models/backbones_3d/pointnet2_backbone.py:56-94 (backbone glue).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cpu_oracle as o


@torch.no_grad()
def sa_forward(sa_cpu, xyz, features):
    """sa_cpu: a PointnetSAModuleMSG on the CPU (weights only are used).  xyz (B,N,3) np, features (B,C,N) np|None."""
    idx = o.furthest_point_sample(xyz, sa_cpu.npoint)
    new_xyz = np.ascontiguousarray(np.take_along_axis(xyz, idx[:, :, None].astype(np.int64), 1))
    outs = []
    for grouper, mlp in zip(sa_cpu.groupers, sa_cpu.mlps):
        grouped, _ = o.query_and_group(grouper.radius, grouper.nsample, xyz, new_xyz, features)
        x = mlp(torch.from_numpy(grouped))
        x = F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
        outs.append(x)
    return new_xyz, torch.cat(outs, dim=1).numpy()


@torch.no_grad()
def fp_forward(fp_cpu, unknown, known, unknow_feats, known_feats):
    dist, idx = o.three_nn(unknown, known)
    d = torch.from_numpy(dist)
    dist_recip = 1.0 / (d + 1e-8)
    weight = (dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)).numpy()
    interp = torch.from_numpy(o.three_interpolate(known_feats, idx, weight))
    x = interp if unknow_feats is None else torch.cat([interp, torch.from_numpy(unknow_feats)], dim=1)
    return fp_cpu.mlp(x.unsqueeze(-1)).squeeze(-1).numpy()


@torch.no_grad()
def backbone_forward(backbone_cpu, clouds):
    """clouds (B,N,3+C) np -> dict(point_features (B*N,Cout), sa_xyz, sa_features)."""
    xyz = np.ascontiguousarray(clouds[:, :, :3])
    feats = np.ascontiguousarray(clouds[:, :, 3:].transpose(0, 2, 1)) if clouds.shape[2] > 3 else None
    l_xyz, l_feat = [xyz], [feats]
    for sa in backbone_cpu.SA_modules:
        nx, nf = sa_forward(sa, l_xyz[-1], l_feat[-1])
        l_xyz.append(nx)
        l_feat.append(nf)
    sa_xyz, sa_feat = list(l_xyz), list(l_feat)
    for i in range(-1, -(len(backbone_cpu.FP_modules) + 1), -1):
        l_feat[i - 1] = fp_forward(backbone_cpu.FP_modules[i], l_xyz[i - 1], l_xyz[i], l_feat[i - 1], l_feat[i])
    pf = np.ascontiguousarray(l_feat[0].transpose(0, 2, 1)).reshape(-1, l_feat[0].shape[1])
    return {'point_features': pf, 'sa_xyz': sa_xyz, 'sa_features': sa_feat}


@torch.no_grad()
def neck_forward(neck_cpu, sa_xyz, sa_features):
    """PDMNeck (eval) on the CPU: torch for the two 1x1 convs, the oracle for scatter + normalise.
    Returns spatial_features as a (B, C*D, H, W) array."""
    xyz = sa_xyz[neck_cpu.source_layer]
    src = torch.from_numpy(sa_features[neck_cpu.source_layer])
    feat = neck_cpu.proj(src).transpose(1, 2).contiguous().numpy()
    co = neck_cpu.coef(src).transpose(1, 2)
    sh = co[..., :neck_cpu.nsh].contiguous().numpy()
    sigma = F.softplus(co[..., neck_cpu.nsh]) + neck_cpu.sigma_min
    inv2s2 = (0.5 / (sigma * sigma)).contiguous().numpy()
    g = neck_cpu.grid
    dims = (g.W, g.H, g.D)
    grid, wsum = o.pdm_scatter(xyz, feat, sh, inv2s2, g.origin, g.cell, g.inv_cell, dims, neck_cpu.dilation,
                               neck_cpu.degree, layout=1)
    if neck_cpu.normalize:
        grid = o.pdm_normalize(grid, wsum, neck_cpu.feature_dim, dims, layout=1)
    return np.ascontiguousarray(grid.transpose(0, 3, 1, 2))
