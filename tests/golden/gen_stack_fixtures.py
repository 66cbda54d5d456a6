#!/usr/bin/env python3
"""Generates tests/golden/ref_stack_modules.npz by running the REFERENCE's own stack modules
(/root/reference/pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py + pointnet2_utils.py, imported from where
they lie, nothing copied) on the CPU, with the native extension `pointnet2_stack_cuda` replaced by a stub backed by
this repo's CPU oracle (oracle/pointnet2_stack_oracle.c) and torch.cuda.FloatTensor / IntTensor replaced by CPU
factories.  The reference's glue (ball query -> empty-ball mask -> grouping -> centre subtraction -> zeroing ->
Conv/BN/ReLU -> max-pool; three_nn -> inverse-distance weights -> interpolate -> concat -> MLP) then produces the
expected outputs for seeded ragged inputs and seeded weights.
Run in the authoring container only (needs /root/reference); the .npz output is committed.
"""
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
    pkg('pcdet.ops.pointnet2.pointnet2_stack', f'{REF}/pcdet/ops/pointnet2/pointnet2_stack')
    ext = types.ModuleType('pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda')

    def n(t):
        return t.detach().numpy()

    def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
        # the raw kernel result: -1 in slot 0 of an empty ball (the reference's python turns it into mask + zeros)
        got, empty = o.stack_ball_query(radius, nsample, n(xyz), n(xyz_batch_cnt), n(new_xyz), n(new_xyz_batch_cnt))
        got = got.copy()
        got[empty, 0] = -1
        idx.copy_(torch.from_numpy(got))

    def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
        out.copy_(torch.from_numpy(o.stack_grouping_operation(n(features), n(features_batch_cnt), n(idx), n(idx_batch_cnt))))

    def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
        d, i = o.stack_three_nn(n(unknown), n(unknown_batch_cnt), n(known), n(known_batch_cnt))
        dist2.copy_(torch.from_numpy(d * d)); idx.copy_(torch.from_numpy(i))

    def three_interpolate_wrapper(features, idx, weight, out):
        out.copy_(torch.from_numpy(o.stack_three_interpolate(n(features), n(idx), n(weight))))

    def stack_farthest_point_sampling_wrapper(xyz, temp, xyz_batch_cnt, idx, num_sampled_points):
        idx.copy_(torch.from_numpy(o.stack_furthest_point_sample(n(xyz), n(xyz_batch_cnt), n(num_sampled_points).tolist())))

    for f in (ball_query_wrapper, group_points_wrapper, three_nn_wrapper, three_interpolate_wrapper,
              stack_farthest_point_sampling_wrapper):
        setattr(ext, f.__name__, f)
    sys.modules[ext.__name__] = ext
    sys.modules['pcdet.ops.pointnet2.pointnet2_stack'].pointnet2_stack_cuda = ext
    torch.cuda.FloatTensor = lambda *s: torch.empty(*s, dtype=torch.float32)
    torch.cuda.IntTensor = lambda *s: torch.empty(*s, dtype=torch.int32)
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules, pointnet2_utils  # the reference's files
    return pointnet2_modules, pointnet2_utils


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
    pm, pu = install_reference()
    rng = np.random.default_rng(77)
    out = {}
    counts, mcounts = [150, 60, 90], [20, 7, 13]
    xyz = rng.uniform(0, 3, (sum(counts), 3)).astype(np.float32)
    starts = np.concatenate([[0], np.cumsum(counts)])
    new_xyz = np.concatenate([xyz[starts[b]:starts[b] + m] for b, m in enumerate(mcounts)]).copy()
    new_xyz[5] += 50.0                                   # an empty ball
    feat = rng.standard_normal((sum(counts), 4)).astype(np.float32)
    xc, nc = torch.tensor(counts, dtype=torch.int32), torch.tensor(mcounts, dtype=torch.int32)

    sa = pm.StackSAModuleMSG(radii=[0.4, 0.8], nsamples=[8, 16], mlps=[[4, 8, 16], [4, 8, 24]], use_xyz=True).eval()
    randomize(sa, 5)
    with torch.no_grad():
        _, nf = sa(torch.from_numpy(xyz), xc, torch.from_numpy(new_xyz), nc, torch.from_numpy(feat))
        grouped, idx = pu.QueryAndGroup(0.4, 8, use_xyz=True)(torch.from_numpy(xyz), xc, torch.from_numpy(new_xyz), nc,
                                                              torch.from_numpy(feat))
    out.update(counts=np.array(counts, np.int32), mcounts=np.array(mcounts, np.int32), xyz=xyz, new_xyz=new_xyz, feat=feat,
               sa_out=nf.numpy(), qg_out=grouped.numpy(), qg_idx=idx.numpy())
    for k, v in sa.state_dict().items():
        out['sa_state.' + k] = v.numpy()

    fp = pm.StackPointnetFPModule(mlp=[16 + 4, 12, 8]).eval()
    randomize(fp, 6)
    kfeat = rng.standard_normal((sum(mcounts), 16)).astype(np.float32)
    with torch.no_grad():
        fo = fp(torch.from_numpy(xyz), xc, torch.from_numpy(new_xyz), nc, torch.from_numpy(feat), torch.from_numpy(kfeat))
        fps = pu.stack_farthest_point_sample(torch.from_numpy(xyz), xc, [10, 5, 8])
    out.update(fp_kfeat=kfeat, fp_out=fo.numpy(), fps_idx=fps.numpy())
    for k, v in fp.state_dict().items():
        out['fp_state.' + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_stack_modules.npz'), **out)
    print('wrote ref_stack_modules.npz:', {k: v.shape for k, v in out.items() if not k.startswith(('sa_state', 'fp_state'))})


if __name__ == '__main__':
    main()
