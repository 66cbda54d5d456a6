"""CPU autograd statement of the TRAINING step of the hot path: PointNet2MSG (train mode, BatchNorm batch
statistics) + PDM neck, fp32, built on the oracle operators.

TEST INFRASTRUCTURE ONLY (same rule as cpu_oracle.py / cpu_backbone.py): the checker of the GPU training path
(tests/test_configs_gpu.py, BASELINE config 4).  Forward values and gradients of the operators come from the C
oracle (pointnet2_oracle.c: group_points / three_interpolate and their scatter-add backward,
/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/group_points_gpu.cu:14-31, interpolate_gpu.cu:127-149;
pdm_oracle.c for the neck's scatter and its gradient); the dense layers are the torch modules under test run
on the CPU in fp32, following pointnet2_modules.py:19-55 (SA) and :141-170 (FP).
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import cpu_oracle as o


class _QueryAndGroup(Function):
    """QueryAndGroup.forward (pointnet2_utils.py:241-264): (B, 3+C, M, ns); gradient to the features only."""

    @staticmethod
    def forward(ctx, features, xyz, new_xyz, radius, nsample):
        grouped, idx = o.query_and_group(radius, nsample, xyz, new_xyz, None if features is None else features.detach().numpy())
        ctx.idx, ctx.n = idx, xyz.shape[1]
        return torch.from_numpy(grouped)

    @staticmethod
    def backward(ctx, g):
        gf = o.grouping_operation_grad(np.ascontiguousarray(g.numpy()[:, 3:]), ctx.idx, ctx.n)
        return torch.from_numpy(gf), None, None, None, None


class _ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, feats, idx, weight):
        ctx.idx, ctx.w, ctx.m = idx, weight, feats.shape[2]
        return torch.from_numpy(o.three_interpolate(feats.detach().numpy(), idx, weight))

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(o.three_interpolate_grad(np.ascontiguousarray(g.numpy()), ctx.idx, ctx.w, ctx.m)), None, None


class _PdmScatter(Function):
    @staticmethod
    def forward(ctx, feat, sh, inv2s2, xyz, spec):
        origin, cell, inv_cell, dims, kernel, degree = spec
        grid, wsum = o.pdm_scatter(xyz, feat.detach().numpy(), sh.detach().numpy(), inv2s2.detach().numpy(), origin, cell, inv_cell, dims, kernel,
                                   degree, layout=1)
        ctx.save_for_backward(feat, sh, inv2s2)
        ctx.xyz, ctx.spec = xyz, spec
        return torch.from_numpy(grid), torch.from_numpy(wsum)

    @staticmethod
    def backward(ctx, dgrid, dwsum):
        feat, sh, inv2s2 = ctx.saved_tensors
        origin, cell, inv_cell, dims, kernel, degree = ctx.spec
        dfeat, dsh, dinv = o.pdm_scatter_grad(ctx.xyz, feat.numpy(), sh.numpy(), inv2s2.numpy(), origin, cell, inv_cell,
                                              dims, kernel, degree, np.ascontiguousarray(dgrid.numpy()),
                                              None if dwsum is None else np.ascontiguousarray(dwsum.numpy()), layout=1)
        return torch.from_numpy(dfeat), torch.from_numpy(dsh), torch.from_numpy(dinv), None, None


def sa_forward(sa, xyz, features):
    """xyz (B,N,3) numpy, features (B,C,N) torch|None -> new_xyz numpy, features (B,Cout,M) torch (with graph)."""
    idx = o.furthest_point_sample(xyz, sa.npoint)
    new_xyz = np.ascontiguousarray(np.take_along_axis(xyz, idx[:, :, None].astype(np.int64), 1))
    outs = []
    for grouper, mlp in zip(sa.groupers, sa.mlps):
        x = mlp(_QueryAndGroup.apply(features, xyz, new_xyz, grouper.radius, grouper.nsample))
        outs.append(F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1))
    return new_xyz, torch.cat(outs, dim=1)


def fp_forward(fp, unknown, known, unknown_feats, known_feats):
    dist, idx = o.three_nn(unknown, known)
    d = torch.from_numpy(dist)
    dist_recip = 1.0 / (d + 1e-8)
    weight = (dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)).numpy()
    interp = _ThreeInterpolate.apply(known_feats, idx, weight)
    x = interp if unknown_feats is None else torch.cat([interp, unknown_feats], dim=1)
    return fp.mlp(x.unsqueeze(-1)).squeeze(-1)


def train_forward(backbone, neck, clouds):
    """backbone, neck: CPU fp32 modules in train() mode; clouds (B,N,3+C) numpy.
    -> dict(point_features (B*N,C) torch, spatial_features (B,C*D,H,W) torch, sa_xyz [numpy])."""
    xyz = np.ascontiguousarray(clouds[:, :, :3])
    feats = torch.from_numpy(np.ascontiguousarray(clouds[:, :, 3:].transpose(0, 2, 1))) if clouds.shape[2] > 3 else None
    l_xyz, l_feat = [xyz], [feats]
    for sa in backbone.SA_modules:
        nx, nf = sa_forward(sa, l_xyz[-1], l_feat[-1])
        l_xyz.append(nx)
        l_feat.append(nf)
    sa_xyz, sa_feat = list(l_xyz), list(l_feat)
    for i in range(-1, -(len(backbone.FP_modules) + 1), -1):
        l_feat[i - 1] = fp_forward(backbone.FP_modules[i], l_xyz[i - 1], l_xyz[i], l_feat[i - 1], l_feat[i])
    pf = l_feat[0].permute(0, 2, 1).reshape(-1, l_feat[0].shape[1])
    out = {'point_features': pf, 'sa_xyz': sa_xyz, 'sa_features': sa_feat}
    if neck is not None:
        src = sa_feat[neck.source_layer]
        feat = neck.proj(src).transpose(1, 2).contiguous()
        co = neck.coef(src).transpose(1, 2)
        sh = co[..., :neck.nsh].contiguous()
        sigma = F.softplus(co[..., neck.nsh]) + neck.sigma_min
        inv2s2 = (0.5 / (sigma * sigma)).contiguous()
        g = neck.grid
        spec = (g.origin, g.cell, g.inv_cell, (g.W, g.H, g.D), neck.dilation, neck.degree)
        grid, wsum = _PdmScatter.apply(feat, sh, inv2s2, sa_xyz[neck.source_layer], spec)   # (B,H,W,C*D), (B,H,W,D)
        if neck.normalize:
            B = grid.shape[0]
            w = wsum.unsqueeze(3)                                                            # (B,H,W,1,D)
            g5 = grid.view(B, g.H, g.W, neck.feature_dim, g.D)
            ok = w.abs() > 1e-6
            grid = torch.where(ok, g5 / torch.where(ok, w, torch.ones_like(w)), g5).reshape(B, g.H, g.W, -1)
        out['spatial_features'] = grid.permute(0, 3, 1, 2)
        out['pdm_weight_sum'] = wsum
    return out
