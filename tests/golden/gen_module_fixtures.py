#!/usr/bin/env python3
"""Generates tests/golden/ref_modules.npz by running the REFERENCE's own torch modules
(/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py + pointnet2_utils.py, imported
from where they lie, nothing copied) on the CPU, with
  - the native extension `pointnet2_batch_cuda` replaced by a stub backed by this repo's CPU oracle, and
  - torch.cuda.FloatTensor / IntTensor replaced by CPU factories (the reference's Functions allocate with them),
so the reference's module glue (FPS -> gather -> QueryAndGroup -> Conv/BN/ReLU -> max-pool; three_nn ->
inverse-distance weights -> interpolate -> concat -> MLP) produces the expected outputs for seeded inputs
and seeded weights.  Also records the reference modules' state_dict key/shape manifest.
Run in the authoring container only (needs /root/reference); the .npz/.json outputs are committed.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import cpu_oracle as o  # noqa: E402

REF = '/root/reference'


def install_reference():
    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m
    pkg('pcdet', f'{REF}/pcdet')
    pkg('pcdet.ops', f'{REF}/pcdet/ops')
    pkg('pcdet.ops.pointnet2', f'{REF}/pcdet/ops/pointnet2')
    pkg('pcdet.ops.pointnet2.pointnet2_batch', f'{REF}/pcdet/ops/pointnet2/pointnet2_batch')

    ext = types.ModuleType('pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda')

    def n(t):
        return t.detach().numpy()

    def ball_query_wrapper(b, nn_, m, radius, nsample, new_xyz, xyz, idx):
        idx.copy_(torch.from_numpy(o.ball_query(radius, nsample, n(xyz), n(new_xyz))))
        return 1

    def group_points_wrapper(b, c, nn_, npoints, nsample, points, idx, out):
        out.copy_(torch.from_numpy(o.grouping_operation(n(points), n(idx))))
        return 1

    def gather_points_wrapper(b, c, nn_, npoints, points, idx, out):
        out.copy_(torch.from_numpy(o.gather_operation(n(points), n(idx))))
        return 1

    def farthest_point_sampling_wrapper(b, nn_, m, points, temp, idx):
        idx.copy_(torch.from_numpy(o.furthest_point_sample(n(points), m)))
        return 1

    def three_nn_wrapper(b, nn_, m, unknown, known, dist2, idx):
        d2, i = o.three_nn_dist2(n(unknown), n(known))
        dist2.copy_(torch.from_numpy(d2)); idx.copy_(torch.from_numpy(i))

    def three_interpolate_wrapper(b, c, m, nn_, points, idx, weight, out):
        out.copy_(torch.from_numpy(o.three_interpolate(n(points), n(idx), n(weight))))

    for f in (ball_query_wrapper, group_points_wrapper, gather_points_wrapper, farthest_point_sampling_wrapper,
              three_nn_wrapper, three_interpolate_wrapper):
        setattr(ext, f.__name__, f)
    sys.modules[ext.__name__] = ext
    sys.modules['pcdet.ops.pointnet2.pointnet2_batch'].pointnet2_batch_cuda = ext

    torch.cuda.FloatTensor = lambda *s: torch.empty(*s, dtype=torch.float32)
    torch.cuda.IntTensor = lambda *s: torch.empty(*s, dtype=torch.int32)
    from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_modules  # the reference's file
    return pointnet2_modules


def randomize(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1) + (1.0 if p.dim() == 1 else 0.0))
        for name, buf in module.named_buffers():
            if name.endswith('running_mean'):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
            elif name.endswith('running_var'):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)


def main():
    pm = install_reference()
    rng = np.random.default_rng(2024)
    out = {}
    manifest = {}

    # ---- SA MSG module (two scales), B=2, N=256, C=3
    sa = pm.PointnetSAModuleMSG(npoint=32, radii=[0.4, 0.8], nsamples=[8, 16],
                                mlps=[[3, 8, 16], [3, 8, 24]], use_xyz=True).eval()
    randomize(sa, 7)
    xyz = rng.uniform(0, 3, (2, 256, 3)).astype(np.float32)
    xyz[:, 200:] = xyz[:, :56]  # duplicated (padded) points, data_processor.py:206-210
    feat = rng.standard_normal((2, 3, 256)).astype(np.float32)
    with torch.no_grad():
        new_xyz, new_feat = sa(torch.from_numpy(xyz), torch.from_numpy(feat))
    out.update(sa_xyz=xyz, sa_feat=feat, sa_new_xyz=new_xyz.numpy(), sa_new_feat=new_feat.numpy())
    for k, v in sa.state_dict().items():
        out['sa_state.' + k] = v.numpy()
    manifest['PointnetSAModuleMSG(npoint=32,radii=[.4,.8],nsamples=[8,16],mlps=[[3,8,16],[3,8,24]])'] = \
        {k: list(v.shape) for k, v in sa.state_dict().items()}

    # ---- FP module
    fp = pm.PointnetFPModule(mlp=[24 + 5, 16, 12]).eval()
    randomize(fp, 9)
    unknown = rng.uniform(0, 3, (2, 100, 3)).astype(np.float32)
    known = np.ascontiguousarray(unknown[:, :20])
    uf = rng.standard_normal((2, 5, 100)).astype(np.float32)
    kf = rng.standard_normal((2, 24, 20)).astype(np.float32)
    with torch.no_grad():
        fo = fp(torch.from_numpy(unknown), torch.from_numpy(known), torch.from_numpy(uf), torch.from_numpy(kf))
    out.update(fp_unknown=unknown, fp_known=known, fp_uf=uf, fp_kf=kf, fp_out=fo.numpy())
    for k, v in fp.state_dict().items():
        out['fp_state.' + k] = v.numpy()
    manifest['PointnetFPModule(mlp=[29,16,12])'] = {k: list(v.shape) for k, v in fp.state_dict().items()}

    # ---- QueryAndGroup / GroupAll (operator-level glue)
    from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as pu
    nx = torch.from_numpy(np.ascontiguousarray(xyz[:, :16]))
    qg = pu.QueryAndGroup(0.5, 8, use_xyz=True)(torch.from_numpy(xyz), nx, torch.from_numpy(feat))
    ga = pu.GroupAll(True)(torch.from_numpy(xyz), None, torch.from_numpy(feat))
    out.update(qg_new_xyz=nx.numpy(), qg_out=qg.numpy(), groupall_out=ga.numpy())

    # ---- PointNet2MSG-style stack: state_dict manifest for the pointrcnn configuration
    specs = [[[1, 16, 16, 32], [1, 32, 32, 64]], [[96, 64, 64, 128], [96, 64, 96, 128]],
             [[256, 128, 196, 256], [256, 128, 196, 256]], [[512, 256, 256, 512], [512, 256, 384, 512]]]
    npts, radii = [4096, 1024, 256, 64], [[0.1, 0.5], [0.5, 1.0], [1.0, 2.0], [2.0, 4.0]]
    stack = {}
    for i, (mlps, np_, rr) in enumerate(zip(specs, npts, radii)):
        m = pm.PointnetSAModuleMSG(npoint=np_, radii=rr, nsamples=[16, 32], mlps=[list(s) for s in mlps])
        for k, v in m.state_dict().items():
            stack[f'SA_modules.{i}.{k}'] = list(v.shape)
    for i, mlp in enumerate([[257, 128, 128], [608, 256, 256], [768, 512, 512], [1536, 512, 512]]):
        m = pm.PointnetFPModule(mlp=mlp)
        for k, v in m.state_dict().items():
            stack[f'FP_modules.{i}.{k}'] = list(v.shape)
    manifest['PointNet2MSG(pointrcnn, input_channels=4)'] = stack

    np.savez_compressed(os.path.join(HERE, 'ref_modules.npz'), **out)
    with open(os.path.join(HERE, 'ref_state_dict_manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print('wrote', len(out), 'arrays;', {k: len(v) for k, v in manifest.items()})


if __name__ == '__main__':
    main()
