#!/usr/bin/env python3
"""Generates tests/golden/ref_vector_pool_modules.npz by running the REFERENCE's own vector-pool / voxel-query python
(/root/reference/pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py, pointnet2_utils.py, voxel_query_utils.py,
imported from where they lie, nothing copied) on the CPU, with the native extension `pointnet2_stack_cuda` replaced
by a stub backed by this repo's CPU oracle (oracle/vector_pool_oracle.c, pointnet2_stack_oracle.c).  What the
reference's python adds on top of the kernels — the buffer-size retry loops, the division by the cell counts, the
inverse-distance weights, the empty-cell masks, the lattice of cell centres, the channel-group sum, the grouped 1x1
convolution and MLPs, the global -> local index arithmetic of VoxelQueryAndGrouping — then produces the expected
outputs for seeded ragged inputs and seeded weights.
Run in the authoring container only (needs /root/reference); the .npz output is committed.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_stack_fixtures as base  # noqa: E402  (installs the stubbed reference package + the PointNet++ stack stubs)

o = base.o


def install_vector_pool_stubs():
    pm, pu = base.install_reference()
    ext = sys.modules['pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda']

    def n(t):
        return t.detach().numpy()

    def voxel_query_wrapper(M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, point_indices, idx):
        got, empty = o.stack_voxel_query((z_range, y_range, x_range), radius, nsample, n(xyz), n(new_xyz), n(new_coords), n(point_indices))
        got = got.copy()
        got[empty, 0] = -1                      # the raw kernel result; the reference's python makes the mask from it
        idx.copy_(torch.from_numpy(got))

    def query_stacked_local_neighbor_idxs_wrapper_stack(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, stack_neighbor_idxs,
                                                        start_len, cumsum, avg_length_of_neighbor_idxs, max_neighbour_distance, nsample,
                                                        neighbor_type):
        stack, sl, total = o.stack_query_local_neighbor_idxs(n(support_xyz), n(xyz_batch_cnt), n(new_xyz), n(new_xyz_batch_cnt),
                                                             avg_length_of_neighbor_idxs, max_neighbour_distance, nsample, neighbor_type)
        stack_neighbor_idxs.copy_(torch.from_numpy(stack)); start_len.copy_(torch.from_numpy(sl)); cumsum.fill_(total)

    def query_three_nn_by_stacked_local_idxs_wrapper_stack(support_xyz, new_xyz, new_xyz_grid_centers, new_xyz_grid_idxs,
                                                           new_xyz_grid_dist2, stack_neighbor_idxs, start_len, M, num_total_grids):
        d2, idx = o.stack_three_nn_by_local_idxs(n(support_xyz), n(new_xyz_grid_centers), n(stack_neighbor_idxs), n(start_len))
        new_xyz_grid_dist2.copy_(torch.from_numpy(d2)); new_xyz_grid_idxs.copy_(torch.from_numpy(idx))

    def vector_pool_wrapper(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, new_features, new_local_xyz,
                            point_cnt_of_grid, grouped_idxs, num_grid_x, num_grid_y, num_grid_z, max_neighbour_distance, use_xyz,
                            num_max_sum_points, nsample, neighbor_type, pooling_type):
        nf, nl, pc, grouped, total = o.stack_vector_pool_once(
            n(support_xyz), n(xyz_batch_cnt), n(support_features), n(new_xyz), n(new_xyz_batch_cnt), (num_grid_x, num_grid_y, num_grid_z),
            max_neighbour_distance, new_features.shape[1], use_xyz, num_max_sum_points, nsample, neighbor_type, pooling_type)
        new_features.copy_(torch.from_numpy(nf)); new_local_xyz.copy_(torch.from_numpy(nl))
        point_cnt_of_grid.copy_(torch.from_numpy(pc)); grouped_idxs.copy_(torch.from_numpy(grouped))
        return total

    def vector_pool_grad_wrapper(grad_new_features, point_cnt_of_grid, grouped_idxs, grad_support_features):
        N, c_in = grad_support_features.shape
        grad_support_features.copy_(torch.from_numpy(o.stack_vector_pool_grad(n(grad_new_features), n(point_cnt_of_grid), n(grouped_idxs), N, c_in)))

    def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
        grad_features.copy_(torch.from_numpy(o.stack_three_interpolate_grad(n(grad_out), n(idx), n(weight), grad_features.shape[0])))

    for f in (voxel_query_wrapper, query_stacked_local_neighbor_idxs_wrapper_stack, query_three_nn_by_stacked_local_idxs_wrapper_stack,
              vector_pool_wrapper, vector_pool_grad_wrapper, three_interpolate_grad_wrapper):
        setattr(ext, f.__name__, f)
    from pcdet.ops.pointnet2.pointnet2_stack import voxel_query_utils  # the reference's file
    return pm, pu, voxel_query_utils


class Cfg(dict):
    """attribute + item access, as the reference's EasyDict configs"""
    __getattr__ = dict.__getitem__


def msg_config(kind):
    return Cfg(NAME='VectorPoolAggregationModuleMSG', NUM_GROUPS=2, LOCAL_AGGREGATION_TYPE=kind, NUM_REDUCED_CHANNELS=4,
               NUM_CHANNELS_OF_LOCAL_AGGREGATION=6, MSG_POST_MLPS=[10],
               GROUP_CFG_0=Cfg(NUM_LOCAL_VOXEL=[2, 2, 2], MAX_NEIGHBOR_DISTANCE=0.5, NEIGHBOR_NSAMPLE=-1, POST_MLPS=[8, 8]),
               GROUP_CFG_1=Cfg(NUM_LOCAL_VOXEL=[3, 3, 2], MAX_NEIGHBOR_DISTANCE=0.9, NEIGHBOR_NSAMPLE=-1, POST_MLPS=[12, 8]))


def main():
    pm, pu, vq = install_vector_pool_stubs()
    rng = np.random.default_rng(123)
    out = {}
    counts, mcounts = [220, 140], [16, 16]
    xyz = rng.uniform(0, 3, (sum(counts), 3)).astype(np.float32)
    starts = np.concatenate([[0], np.cumsum(counts)])
    new_xyz = np.concatenate([xyz[starts[b]:starts[b] + m] + rng.normal(0, 0.05, (m, 3)).astype(np.float32) for b, m in enumerate(mcounts)])
    new_xyz = new_xyz.astype(np.float32)
    new_xyz[3] += 40.0                                  # a key point with no neighbours
    feat = rng.standard_normal((sum(counts), 8)).astype(np.float32)
    xc, nc = torch.tensor(counts, dtype=torch.int32), torch.tensor(mcounts, dtype=torch.int32)
    out.update(counts=np.array(counts, np.int32), mcounts=np.array(mcounts, np.int32), xyz=xyz, new_xyz=new_xyz, feat=feat)

    for tag, kind in (('interp', 'local_interpolation'), ('avg', 'voxel_avg_pool'), ('first', 'voxel_random_choice')):
        layer, c_out = pm.build_local_aggregation_module(8, msg_config(kind))
        layer.eval()
        base.randomize(layer, 11)
        f = torch.from_numpy(feat).requires_grad_(True)
        key, nf = layer(xyz=torch.from_numpy(xyz), xyz_batch_cnt=xc, new_xyz=torch.from_numpy(new_xyz), new_xyz_batch_cnt=nc, features=f)
        g = torch.from_numpy(rng.standard_normal(tuple(nf.shape)).astype(np.float32))
        nf.backward(g)
        out[f'{tag}_out'], out[f'{tag}_grad_out'], out[f'{tag}_grad_feat'] = nf.detach().numpy(), g.numpy(), f.grad.numpy()
        assert c_out == 10 and torch.equal(key, torch.from_numpy(new_xyz))
        for k, v in layer.state_dict().items():
            out[f'{tag}_state.' + k] = v.numpy()
        # the gather stage of the second scale on its own (what the kernels + the reference's python hand to the MLPs)
        sub = layer.layer_1
        red = torch.from_numpy(feat).view(feat.shape[0], -1, 4).sum(1)
        with torch.no_grad():
            if kind == 'local_interpolation':
                vec = sub.vector_pool_with_local_interpolate(torch.from_numpy(xyz), xc, red, torch.from_numpy(new_xyz), nc)
            else:
                vec, cnt = sub.vector_pool_with_voxel_query(torch.from_numpy(xyz), xc, red.contiguous(), torch.from_numpy(new_xyz), nc)
                out[f'{tag}_cnt'] = cnt.numpy()
        out[f'{tag}_vec'] = vec.numpy()

    # VoxelQueryAndGrouping: one point per voxel (the last written), equal key-point counts per sample (ref :85)
    voxel, lo = np.float32(0.5), np.zeros(3, np.float32)
    Z = Y = X = 7
    vox = np.full((2, Z, Y, X), -1, np.int32)
    for b in range(2):
        c = np.floor((xyz[starts[b]:starts[b + 1]] - lo) / voxel).astype(np.int64)
        for i in range(counts[b]):
            vox[b, c[i, 2], c[i, 1], c[i, 0]] = starts[b] + i
    kc = np.floor((new_xyz - lo) / voxel).astype(np.int64)
    coords = np.stack([np.repeat([0, 1], 16), kc[:, 2], kc[:, 1], kc[:, 0]], 1).astype(np.int32)
    mod = vq.VoxelQueryAndGrouping((1, 2, 2), 0.8, 6)
    with torch.no_grad():
        gf, gx, mask = mod(torch.from_numpy(coords), torch.from_numpy(xyz), xc, torch.from_numpy(new_xyz), nc, torch.from_numpy(feat),
                           torch.from_numpy(vox))
    out.update(vq_vox=vox, vq_coords=coords, vq_feat=gf.numpy(), vq_xyz=gx.numpy(), vq_mask=mask.numpy())
    assert mask[3] and not mask.all()

    np.savez_compressed(os.path.join(HERE, 'ref_vector_pool_modules.npz'), **out)
    print('wrote ref_vector_pool_modules.npz:', {k: v.shape for k, v in out.items() if '_state.' not in k})


if __name__ == '__main__':
    main()
