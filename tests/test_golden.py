"""Oracle and module restatement against the committed fixtures under tests/golden/."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_backbone
from oracle import cpu_oracle as o

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(G, "oracle_vectors.npz"))


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(G, "ref_modules.npz"))


def test_oracle_regression_vectors(vec):
    xyz, feat = vec['xyz'], vec['feat']
    fidx = o.furthest_point_sample(xyz, 128)
    np.testing.assert_array_equal(fidx, vec['fps_idx'])
    new_xyz = np.ascontiguousarray(np.take_along_axis(xyz, fidx[:, :, None].astype(np.int64), 1))
    np.testing.assert_array_equal(o.gather_operation(np.ascontiguousarray(xyz.transpose(0, 2, 1)), fidx), vec['gather_xyz'])
    for r, ns in [(0.5, 16), (1.0, 32), (2.0, 16), (4.0, 32)]:
        np.testing.assert_array_equal(o.ball_query(r, ns, xyz, new_xyz), vec[f'ball_r{r}_ns{ns}'])
    idx = vec['ball_r1.0_ns32']
    np.testing.assert_array_equal(o.grouping_operation(feat, idx), vec['group_feat'])
    np.testing.assert_array_equal(o.grouping_operation_grad(vec['group_grad_in'], idx, 1024), vec['group_grad'])
    d2, i3 = o.three_nn_dist2(xyz, new_xyz)
    np.testing.assert_array_equal(i3, vec['nn_idx']); np.testing.assert_array_equal(d2, vec['nn_dist2'])
    np.testing.assert_array_equal(o.three_interpolate(vec['interp_feat'], i3, vec['interp_w']), vec['interp_out'])
    np.testing.assert_array_equal(o.three_interpolate_grad(vec['interp_grad_in'], i3, vec['interp_w'], 128), vec['interp_grad'])


def test_pdm_regression_vectors(vec):
    origin, cell, inv_cell, dims = o.pdm_grid_params((0, -40, -3, 70.4, 40, 1), (1.6, 1.6, 2.0))
    g, ws = o.pdm_scatter(vec['pdm_xyz'], vec['pdm_feat'], vec['pdm_sh'], vec['pdm_is2'], origin, cell, inv_cell,
                          dims, (5, 5, 3), 2, layout=1)
    assert tuple(g.shape) == tuple(vec['pdm_grid_shape'])
    nz = np.flatnonzero(g)
    np.testing.assert_array_equal(nz, vec['pdm_nz_index'])
    np.testing.assert_allclose(g.reshape(-1)[nz], vec['pdm_nz_value'], rtol=1e-6, atol=1e-7)
    assert abs(float(ws.sum()) - float(vec['pdm_wsum_sum'])) < 1e-3
    # layout 0 holds the same numbers in (B, C*D, H, W) order
    g0, _ = o.pdm_scatter(vec['pdm_xyz'], vec['pdm_feat'], vec['pdm_sh'], vec['pdm_is2'], origin, cell, inv_cell,
                          dims, (5, 5, 3), 2, layout=0)
    np.testing.assert_array_equal(g0, g.transpose(0, 3, 1, 2))


def _load(module, ref, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(ref[k]) for k in ref.files if k.startswith(prefix)}
    missing = module.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return module.eval()


def test_sa_module_matches_reference_module(ref):
    """The reference's PointnetSAModuleMSG.forward (run through gen_module_fixtures.py) == this repo's
    module weights + CPU statement of the SA graph; state_dict loads with strict=True (same key names)."""
    from pdm_ssd_amd.pointnet2_batch import pointnet2_modules as pm
    sa = pm.PointnetSAModuleMSG(npoint=32, radii=[0.4, 0.8], nsamples=[8, 16], mlps=[[3, 8, 16], [3, 8, 24]])
    _load(sa, ref, 'sa_state.')
    new_xyz, new_feat = cpu_backbone.sa_forward(sa, ref['sa_xyz'], ref['sa_feat'])
    np.testing.assert_array_equal(new_xyz, ref['sa_new_xyz'])
    np.testing.assert_allclose(new_feat, ref['sa_new_feat'], rtol=1e-5, atol=1e-5)


def test_fp_module_matches_reference_module(ref):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_modules as pm
    fp = _load(pm.PointnetFPModule(mlp=[29, 16, 12]), ref, 'fp_state.')
    out = cpu_backbone.fp_forward(fp, ref['fp_unknown'], ref['fp_known'], ref['fp_uf'], ref['fp_kf'])
    np.testing.assert_allclose(out, ref['fp_out'], rtol=1e-5, atol=1e-5)


def test_query_and_group_matches_reference_glue(ref):
    out, _ = o.query_and_group(0.5, 8, ref['sa_xyz'], ref['qg_new_xyz'], ref['sa_feat'])
    np.testing.assert_array_equal(out, ref['qg_out'])
    ga = np.concatenate([ref['sa_xyz'].transpose(0, 2, 1)[:, :, None], ref['sa_feat'][:, :, None]], 1)
    np.testing.assert_array_equal(ga, ref['groupall_out'])


def test_backbone_state_dict_keys_match_reference():
    from pdm_ssd_amd.pointnet2_backbone import POINTRCNN_MSG_CFG, PointNet2MSG
    manifest = json.load(open(os.path.join(G, "ref_state_dict_manifest.json")))
    want = manifest['PointNet2MSG(pointrcnn, input_channels=4)']
    got = {k: list(v.shape) for k, v in PointNet2MSG(POINTRCNN_MSG_CFG, input_channels=4).state_dict().items()}
    assert got == want


@pytest.mark.gpu
def test_gpu_modules_match_reference_fixtures(ref, dev):
    """Same fixtures, HIP path: unfused autograd graph and fused MFMA kernels."""
    from pdm_ssd_amd.pointnet2_batch import pointnet2_modules as pm
    from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    sa = pm.PointnetSAModuleMSG(npoint=32, radii=[0.4, 0.8], nsamples=[8, 16], mlps=[[3, 8, 16], [3, 8, 24]])
    sa = _load(sa, ref, 'sa_state.').to(dev)
    with torch.no_grad():
        nx, nf = sa(t(ref['sa_xyz']), t(ref['sa_feat']))
    np.testing.assert_array_equal(nx.cpu().numpy(), ref['sa_new_xyz'])
    np.testing.assert_allclose(nf.cpu().numpy(), ref['sa_new_feat'], rtol=1e-4, atol=1e-4)
    fp = _load(pm.PointnetFPModule(mlp=[29, 16, 12]), ref, 'fp_state.').to(dev)
    with torch.no_grad():
        fo = fp(t(ref['fp_unknown']), t(ref['fp_known']), t(ref['fp_uf']), t(ref['fp_kf']))
    np.testing.assert_allclose(fo.cpu().numpy(), ref['fp_out'], rtol=1e-4, atol=1e-4)
    qg = pu.QueryAndGroup(0.5, 8)(t(ref['sa_xyz']), t(ref['qg_new_xyz']), t(ref['sa_feat']))
    np.testing.assert_array_equal(qg.cpu().numpy(), ref['qg_out'])
    ga = pu.GroupAll(True)(t(ref['sa_xyz']), None, t(ref['sa_feat']))
    np.testing.assert_array_equal(ga.cpu().numpy(), ref['groupall_out'])


@pytest.mark.gpu
def test_gpu_ops_match_regression_vectors(vec, dev):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xyz = t(vec['xyz'])
    fidx = pu.furthest_point_sample(xyz, 128)
    np.testing.assert_array_equal(fidx.cpu().numpy(), vec['fps_idx'])
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), fidx).transpose(1, 2).contiguous()
    for r, ns in [(0.5, 16), (1.0, 32), (2.0, 16), (4.0, 32)]:
        np.testing.assert_array_equal(pu.ball_query(r, ns, xyz, new_xyz).cpu().numpy(), vec[f'ball_r{r}_ns{ns}'])
    np.testing.assert_array_equal(pu.grouping_operation(t(vec['feat']), t(vec['ball_r1.0_ns32'])).cpu().numpy(), vec['group_feat'])
    d, i3 = pu.three_nn(xyz, new_xyz)
    np.testing.assert_array_equal(i3.cpu().numpy(), vec['nn_idx'])
    np.testing.assert_array_equal(pu.three_interpolate(t(vec['interp_feat']), i3, t(vec['interp_w'])).cpu().numpy(), vec['interp_out'])


# ---- pointnet2_stack: fixtures produced by the reference's own stack modules (tests/golden/gen_stack_fixtures.py) ----

@pytest.fixture(scope="module")
def sref():
    return np.load(os.path.join(G, "ref_stack_modules.npz"))


def test_stack_query_and_group_matches_reference_glue(sref):
    """The reference's stack QueryAndGroup.forward (mask, centre subtraction, zeroing of empty balls, concat) ==
    the same composition of this repo's stack oracle."""
    idx, empty = o.stack_ball_query(0.4, 8, sref['xyz'], sref['counts'], sref['new_xyz'], sref['mcounts'])
    np.testing.assert_array_equal(idx, sref['qg_idx'])
    gx = o.stack_grouping_operation(sref['xyz'], sref['counts'], idx, sref['mcounts']) - sref['new_xyz'][:, :, None]
    gf = o.stack_grouping_operation(sref['feat'], sref['counts'], idx, sref['mcounts'])
    gx[empty] = 0; gf[empty] = 0
    np.testing.assert_array_equal(np.concatenate([gx, gf], 1), sref['qg_out'])
    assert empty[5] and (sref['qg_out'][5] == 0).all()


def test_stack_module_state_dicts_load_strictly(sref):
    from pdm_ssd_amd.pointnet2_stack import pointnet2_modules as sm
    _load(sm.StackSAModuleMSG(radii=[0.4, 0.8], nsamples=[8, 16], mlps=[[4, 8, 16], [4, 8, 24]]), sref, 'sa_state.')
    _load(sm.StackPointnetFPModule(mlp=[20, 12, 8]), sref, 'fp_state.')


@pytest.mark.gpu
def test_gpu_stack_modules_match_reference_fixtures(sref, dev):
    """HIP stack operators + this repo's stack modules (fused MFMA path and torch path) against the outputs of the
    reference's StackSAModuleMSG / StackPointnetFPModule / QueryAndGroup / stack_farthest_point_sample."""
    from pdm_ssd_amd.pointnet2_stack import pointnet2_modules as sm
    from pdm_ssd_amd.pointnet2_stack import pointnet2_utils as su
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xyz, new_xyz, feat, xc, nc = t(sref['xyz']), t(sref['new_xyz']), t(sref['feat']), t(sref['counts']), t(sref['mcounts'])
    grouped, idx = su.QueryAndGroup(0.4, 8, use_xyz=True)(xyz, xc, new_xyz, nc, feat)
    np.testing.assert_array_equal(idx.cpu().numpy(), sref['qg_idx'])
    np.testing.assert_array_equal(grouped.cpu().numpy(), sref['qg_out'])
    np.testing.assert_array_equal(su.stack_farthest_point_sample(xyz, xc, [10, 5, 8]).cpu().numpy(), sref['fps_idx'])
    sa = _load(sm.StackSAModuleMSG(radii=[0.4, 0.8], nsamples=[8, 16], mlps=[[4, 8, 16], [4, 8, 24]]), sref, 'sa_state.').to(dev)
    fp = _load(sm.StackPointnetFPModule(mlp=[20, 12, 8]), sref, 'fp_state.').to(dev)
    for fused_path in (True, False):
        sa.use_fused = fp.use_fused = fused_path
        with torch.no_grad():
            _, nf = sa(xyz, xc, new_xyz, nc, feat)
            fo = fp(xyz, xc, new_xyz, nc, feat, t(sref['fp_kfeat']))
        np.testing.assert_allclose(nf.cpu().numpy(), sref['sa_out'], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(fo.cpu().numpy(), sref['fp_out'], rtol=1e-4, atol=1e-4)


# ---- vector pool / voxel query: fixtures produced by the reference's own python (tests/golden/gen_vector_pool_fixtures.py) ----

@pytest.fixture(scope="module")
def vref():
    return np.load(os.path.join(G, "ref_vector_pool_modules.npz"))


def _vp_cfg(kind):
    return {'NAME': 'VectorPoolAggregationModuleMSG', 'NUM_GROUPS': 2, 'LOCAL_AGGREGATION_TYPE': kind, 'NUM_REDUCED_CHANNELS': 4,
            'NUM_CHANNELS_OF_LOCAL_AGGREGATION': 6, 'MSG_POST_MLPS': [10],
            'GROUP_CFG_0': {'NUM_LOCAL_VOXEL': [2, 2, 2], 'MAX_NEIGHBOR_DISTANCE': 0.5, 'NEIGHBOR_NSAMPLE': -1, 'POST_MLPS': [8, 8]},
            'GROUP_CFG_1': {'NUM_LOCAL_VOXEL': [3, 3, 2], 'MAX_NEIGHBOR_DISTANCE': 0.9, 'NEIGHBOR_NSAMPLE': -1, 'POST_MLPS': [12, 8]}}


def test_vector_pool_gather_stage_matches_reference_python(vref):
    """What the reference's python makes of the kernels' raw outputs (retry loops, division by the cell counts,
    [mean offset | pooled features] per cell; cell-centre lattice, inverse-distance weights, empty-cell zeroing,
    [interpolated | 3 x offset] per cell) == the same composition of this repo's oracle wrappers."""
    xyz, new, counts, mcounts = vref['xyz'], vref['new_xyz'], vref['counts'], vref['mcounts']
    red = vref['feat'].reshape(-1, 2, 4).sum(1)
    for tag, pooling in (('avg', 0), ('first', 1)):
        r = o.stack_vector_pool(xyz, counts, red, new, mcounts, (3, 3, 2), 0.9, 4, True, 20, -1, 0, pooling)
        vec = np.concatenate([r['new_local_xyz'].reshape(32, 18, 3), r['new_features'].reshape(32, 18, 4)], -1).reshape(32, -1)
        np.testing.assert_array_equal(r['point_cnt_of_grid'], vref[f'{tag}_cnt'])
        np.testing.assert_allclose(vec, vref[f'{tag}_vec'], rtol=0, atol=1e-6)       # torch summed the channel groups
    R, n = 0.9, (3, 3, 2)
    axes = [(-R + (2 * np.arange(k) + 1) * R / k).astype(np.float32) for k in n]
    centres = (new[:, None, :] + np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 3)[None]).astype(np.float32)
    d, i, _ = o.stack_three_nn_for_vector_pool_by_two_step(xyz, counts, new, centres, mcounts, R, -1, 0, 1000, 18, 2.0)
    d, i = d.reshape(-1, 3), i.reshape(-1, 3).copy()
    empty = i[:, 0] == -1
    i[empty] = 0
    with np.errstate(over="ignore", invalid="ignore"):
        rec = np.float32(1.0) / (d + np.float32(1e-8))
        w = (rec / np.maximum(rec.sum(-1, keepdims=True), np.float32(1e-8))).astype(np.float32)
    w[empty] = 0
    vec = np.concatenate([o.stack_three_interpolate(red, i, w), (centres.reshape(-1, 1, 3) - xyz[i]).reshape(-1, 9)], 1)
    vec[empty] = 0
    assert empty.reshape(32, 18)[3].all()
    np.testing.assert_allclose(vec.reshape(32, -1), vref['interp_vec'], rtol=1e-5, atol=1e-5)


def test_vector_pool_msg_state_dict_loads_strictly(vref):
    from pdm_ssd_amd.pointnet2_stack import pointnet2_modules as sm
    for tag, kind in (('interp', 'local_interpolation'), ('avg', 'voxel_avg_pool'), ('first', 'voxel_random_choice')):
        layer, c_out = sm.build_local_aggregation_module(8, _vp_cfg(kind))
        _load(layer, vref, f'{tag}_state.')
        assert c_out == 10


@pytest.mark.gpu
def test_gpu_vector_pool_modules_match_reference_fixtures(vref, dev):
    """HIP vector-pool / voxel-query operators + this repo's modules against the outputs AND input gradients of the
    reference's VectorPoolAggregationModuleMSG (three aggregation types) and VoxelQueryAndGrouping."""
    from pdm_ssd_amd.pointnet2_stack import pointnet2_modules as sm
    from pdm_ssd_amd.pointnet2_stack import voxel_query_utils as vq
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xyz, new_xyz, xc, nc = t(vref['xyz']), t(vref['new_xyz']), t(vref['counts']), t(vref['mcounts'])
    for tag, kind in (('interp', 'local_interpolation'), ('avg', 'voxel_avg_pool'), ('first', 'voxel_random_choice')):
        layer, _ = sm.build_local_aggregation_module(8, _vp_cfg(kind))
        layer = _load(layer, vref, f'{tag}_state.').to(dev).eval()
        f = t(vref['feat']).requires_grad_(True)
        key, out = layer(xyz=xyz, xyz_batch_cnt=xc, new_xyz=new_xyz, new_xyz_batch_cnt=nc, features=f)
        np.testing.assert_allclose(out.detach().cpu().numpy(), vref[f'{tag}_out'], rtol=1e-4, atol=1e-4)
        out.backward(t(vref[f'{tag}_grad_out']))
        np.testing.assert_allclose(f.grad.cpu().numpy(), vref[f'{tag}_grad_feat'], rtol=1e-4, atol=1e-4)
        red = t(vref['feat']).view(-1, 2, 4).sum(1)
        with torch.no_grad():
            if kind == 'local_interpolation':
                vec = layer.layer_1.vector_pool_with_local_interpolate(xyz, xc, red, new_xyz, nc)
            else:
                vec, cnt = layer.layer_1.vector_pool_with_voxel_query(xyz, xc, red.contiguous(), new_xyz, nc)
                np.testing.assert_array_equal(cnt.cpu().numpy(), vref[f'{tag}_cnt'])
        np.testing.assert_allclose(vec.cpu().numpy(), vref[f'{tag}_vec'], rtol=1e-5, atol=1e-5)
    mod = vq.VoxelQueryAndGrouping((1, 2, 2), 0.8, 6)
    gf, gx, mask = mod(t(vref['vq_coords']), xyz, xc, new_xyz, nc, t(vref['feat']), t(vref['vq_vox']))
    np.testing.assert_array_equal(mask.cpu().numpy(), vref['vq_mask'])
    np.testing.assert_array_equal(gf.cpu().numpy(), vref['vq_feat'])
    np.testing.assert_array_equal(gx.cpu().numpy(), vref['vq_xyz'])
