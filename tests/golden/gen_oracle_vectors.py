#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.npz: seeded inputs and the CPU oracle's outputs for the nine
pointnet2_batch operators and the PDM scatter.  These are REGRESSION vectors of this repository's oracle
(the reference has no fixtures of its own and its CUDA kernels cannot run here): they pin
the oracle against accidental change and give the GPU tests a second, file-based source of expected
values.  Hand-derived known answers live in tests/test_oracle_kat.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cpu_oracle as o  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    out = {}
    B, N, M = 2, 1024, 128
    xyz = np.stack([rng.uniform(0, 20, (B, N)), rng.uniform(-10, 10, (B, N)), rng.uniform(-3, 1, (B, N))], -1)
    xyz = xyz.astype(np.float32)
    xyz[:, 900:] = xyz[:, :124]  # padded duplicates
    feat = rng.standard_normal((B, 5, N)).astype(np.float32)
    out['xyz'], out['feat'] = xyz, feat
    fidx = o.furthest_point_sample(xyz, M)
    out['fps_idx'] = fidx
    new_xyz = np.ascontiguousarray(np.take_along_axis(xyz, fidx[:, :, None].astype(np.int64), 1))
    out['gather_xyz'] = o.gather_operation(np.ascontiguousarray(xyz.transpose(0, 2, 1)), fidx)
    for r, ns in [(0.5, 16), (1.0, 32), (2.0, 16), (4.0, 32)]:
        out[f'ball_r{r}_ns{ns}'] = o.ball_query(r, ns, xyz, new_xyz)
    idx = out['ball_r1.0_ns32']
    out['group_feat'] = o.grouping_operation(feat, idx)
    go = rng.standard_normal(out['group_feat'].shape).astype(np.float32)
    out['group_grad_in'] = go
    out['group_grad'] = o.grouping_operation_grad(go, idx, N)
    d2, i3 = o.three_nn_dist2(xyz, new_xyz)
    out['nn_dist2'], out['nn_idx'] = d2, i3
    w = rng.uniform(0.1, 1, (B, N, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    kf = rng.standard_normal((B, 7, M)).astype(np.float32)
    out['interp_w'], out['interp_feat'] = w, kf
    out['interp_out'] = o.three_interpolate(kf, i3, w)
    gi = rng.standard_normal(out['interp_out'].shape).astype(np.float32)
    out['interp_grad_in'] = gi
    out['interp_grad'] = o.three_interpolate_grad(gi, i3, w, M)
    # PDM scatter
    P, C, deg = 64, 12, 2
    pxyz = np.stack([rng.uniform(0, 70.4, (B, P)), rng.uniform(-40, 40, (B, P)), rng.uniform(-3, 1, (B, P))], -1).astype(np.float32)
    pf = rng.standard_normal((B, P, C)).astype(np.float32)
    sh = (rng.standard_normal((B, P, 9)) * 0.5).astype(np.float32); sh[..., 0] += 3
    is2 = (0.5 / rng.uniform(0.3, 1.5, (B, P)) ** 2).astype(np.float32)
    origin, cell, inv_cell, dims = o.pdm_grid_params((0, -40, -3, 70.4, 40, 1), (1.6, 1.6, 2.0))
    g, ws = o.pdm_scatter(pxyz, pf, sh, is2, origin, cell, inv_cell, dims, (5, 5, 3), deg, layout=1)
    nz = np.flatnonzero(g)
    out.update(pdm_xyz=pxyz, pdm_feat=pf, pdm_sh=sh, pdm_is2=is2, pdm_nz_index=nz.astype(np.int64),
               pdm_nz_value=g.reshape(-1)[nz], pdm_wsum_sum=np.float64(ws.sum()), pdm_grid_shape=np.array(g.shape))
    np.savez_compressed(os.path.join(HERE, 'oracle_vectors.npz'), **out)
    print('wrote', len(out), 'arrays')


if __name__ == '__main__':
    main()
