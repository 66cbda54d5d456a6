"""Voxel query and the vector-pool family on the GPU: the HIP kernels through the
reference-shaped python API against oracle/vector_pool_oracle.c.  Indices, counts and the pooled sums are bit-exact
(same candidate order, same accumulation order); the backward adds with float atomics (1e-5)."""
import numpy as np
import pytest
import torch

from pdm_ssd_amd.pointnet2_stack import pointnet2_stack_hip as ext
from pdm_ssd_amd.pointnet2_stack import pointnet2_utils as su
from pdm_ssd_amd.pointnet2_stack import voxel_query_utils as vq
from test_stack_gpu import T, ragged_clouds

pytestmark = pytest.mark.gpu


def centres_from(xyz, counts, mcounts, seed, jitter=0.05):
    rng = np.random.default_rng(seed)
    starts = np.concatenate([[0], np.cumsum(counts)])
    out = []
    for b, m in enumerate(mcounts):
        if m:
            pick = rng.integers(0, max(counts[b], 1), m)
            out.append(xyz[starts[b]:starts[b + 1]][pick] + rng.normal(0, jitter, (m, 3)).astype(np.float32))
    return np.ascontiguousarray(np.concatenate(out).astype(np.float32)) if out else np.zeros((0, 3), np.float32)


def voxelize(xyz, counts, voxel, origin, dims, seed):
    """(B, Z, Y, X) index of ONE point per voxel (the last one written, as a scatter would leave it) or -1."""
    Z, Y, X = dims
    vox = np.full((len(counts), Z, Y, X), -1, np.int32)
    starts = np.concatenate([[0], np.cumsum(counts)])
    for b in range(len(counts)):
        p = xyz[starts[b]:starts[b + 1]]
        c = np.floor((p - origin) / voxel).astype(np.int64)
        ok = ((c >= 0) & (c < np.array([X, Y, Z]))).all(1)
        for i in np.flatnonzero(ok):
            vox[b, c[i, 2], c[i, 1], c[i, 0]] = starts[b] + i
    return vox


@pytest.mark.parametrize("max_range,radius,nsample", [((1, 2, 2), 0.8, 16), ((0, 1, 1), 0.4, 4), ((2, 4, 4), 2.0, 32), ((1, 1, 1), 0.0, 3)])
def test_voxel_query_matches_oracle(oracle, dev, max_range, radius, nsample):
    counts, m = [3000, 1800], 150
    xyz = ragged_clouds(counts, 21, "uniform") * np.float32(0.15)                   # ~40 points per square metre
    lo = xyz.min(0)
    voxel = np.array([0.4, 0.4, 0.25], np.float32)
    dims = tuple(int(v) for v in (np.ceil((xyz.max(0) - lo) / voxel).astype(int) + 1)[::-1])
    vox = voxelize(xyz, counts, voxel, lo, dims, 0)
    new = centres_from(xyz, counts, [m, m], 4)
    new[5] += 500.0
    c = np.floor((new - lo) / voxel).astype(np.int32)
    coords = np.stack([np.repeat([0, 1], m), c[:, 2], c[:, 1], c[:, 0]], 1).astype(np.int32)
    coords[5, 1:] = [dims[0] + 3, dims[1] + 3, dims[2] + 3]                         # a window entirely outside the volume
    ridx, rmask = oracle.stack_voxel_query(max_range, radius, nsample, xyz, new, coords, vox)
    idx, mask = vq.voxel_query(max_range, radius, nsample, T(xyz, dev), T(new, dev), T(coords, dev), T(vox, dev))
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(mask.cpu().numpy(), rmask)
    assert rmask[5] and (radius == 0.0 or not rmask.all())


def test_voxel_query_and_grouping_module(oracle, dev):
    counts, m = [2500, 2500], 64
    xyz = ragged_clouds(counts, 33, "uniform") * np.float32(0.15)
    feats = np.random.default_rng(0).normal(size=(5000, 6)).astype(np.float32)
    lo, voxel = xyz.min(0), np.array([0.5, 0.5, 0.5], np.float32)
    dims = tuple(int(v) for v in (np.ceil((xyz.max(0) - lo) / voxel).astype(int) + 1)[::-1])
    vox = voxelize(xyz, counts, voxel, lo, dims, 0)
    new = centres_from(xyz, counts, [m, m], 9)
    c = np.clip(np.floor((new - lo) / voxel).astype(np.int32), 0, None)
    coords = np.stack([np.repeat([0, 1], m), c[:, 2], c[:, 1], c[:, 0]], 1).astype(np.int32)
    mod = vq.VoxelQueryAndGrouping((1, 2, 2), 1.0, 8)
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array([m, m], np.int32), dev)
    gf, gx, mask = mod(T(coords, dev), T(xyz, dev), xc, T(new, dev), nc, T(feats, dev), T(vox, dev))
    ridx, rmask = oracle.stack_voxel_query((1, 2, 2), 1.0, 8, xyz, new, coords, vox)   # global indices, zero where empty
    np.testing.assert_array_equal(mask.cpu().numpy(), rmask)
    local = ridx - np.repeat([0, counts[0]], m)[:, None]
    local[rmask] = 0
    want_f = oracle.stack_grouping_operation(feats, counts, local.astype(np.int32), [m, m])
    want_x = oracle.stack_grouping_operation(xyz, counts, local.astype(np.int32), [m, m])
    np.testing.assert_array_equal(gf.cpu().numpy(), want_f)
    np.testing.assert_array_equal(gx.cpu().numpy(), want_x)


LOCAL_CASES = [([1500, 0, 700, 90], [100, 0, 37, 90], 0.6, -1, 1), ([1500, 0, 700, 90], [100, 0, 37, 90], 0.6, 5, 0),
               ([4000], [300], 4.0, -1, 1), ([64, 64], [1, 1], 0.0, -1, 0)]


@pytest.mark.parametrize("counts,mcounts,dist,nsample,ntype", LOCAL_CASES)
def test_stacked_local_neighbors_one_call_form(oracle, dev, counts, mcounts, dist, nsample, ntype):
    """query_stacked_local_neighbor_idxs_wrapper_stack with upstream's protocol: exact lists when the stack is large
    enough, the same cut when it is not."""
    xyz = ragged_clouds(counts, 5)
    new = centres_from(xyz, counts, mcounts, 2)
    M = new.shape[0]
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    for avg in (1, 1200):
        rstack, rsl, rtotal = oracle.stack_query_local_neighbor_idxs(xyz, counts, new, mcounts, avg, dist, nsample, ntype)
        stack = torch.zeros((avg * M,), dtype=torch.int32, device=dev)
        sl = torch.zeros((M, 2), dtype=torch.int32, device=dev)
        cum = torch.zeros((1,), dtype=torch.int32, device=dev)
        ext.query_stacked_local_neighbor_idxs_wrapper_stack(T(xyz, dev), xc, T(new, dev), nc, stack, sl, cum, avg, dist, nsample, ntype)
        assert int(cum.item()) == rtotal
        np.testing.assert_array_equal(sl.cpu().numpy(), rsl)
        np.testing.assert_array_equal(stack.cpu().numpy(), rstack)
    if dist >= 3:
        assert rsl[:, 1].max() == 1000                                                # the 1000-neighbour ceiling is exercised


@pytest.mark.parametrize("ntype,nsample,G", [(1, -1, 27), (0, 24, 8), (1, 2, 100)])
def test_three_nn_for_vector_pool_by_two_step(oracle, dev, ntype, nsample, G):
    counts, mcounts = [2500, 1200, 0, 5], [120, 60, 0, 5]
    xyz = ragged_clouds(counts, 14)
    new = centres_from(xyz, counts, mcounts, 3)
    new[7] += 300.0                                                                   # a centre with no neighbours at all
    rng = np.random.default_rng(1)
    centres = (new[:, None, :] + rng.uniform(-0.8, 0.8, (new.shape[0], G, 3))).astype(np.float32)
    rd, ri, ravg = oracle.stack_three_nn_for_vector_pool_by_two_step(xyz, counts, new, centres, mcounts, 0.8, nsample, ntype, 3, G, 1.5)
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    d, i, avg = su.three_nn_for_vector_pool_by_two_step(T(xyz, dev), xc, T(new, dev), T(centres, dev), nc, 0.8, nsample, ntype, 3, G, 1.5)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(d.cpu().numpy(), rd)
    assert int(avg) == ravg and (ri[7] == -1).all() and np.isinf(rd[7]).all()


POOL_CASES = [
    # counts, mcounts, grid, dist, c_in, ceg, use_xyz, nsample, ntype, pooling
    ([1500, 0, 700, 90], [100, 0, 37, 90], (3, 3, 3), 0.8, 16, 16, True, -1, 0, 0),
    ([1500, 0, 700, 90], [100, 0, 37, 90], (3, 3, 3), 0.8, 32, 8, True, -1, 1, 0),
    ([1500, 0, 700, 90], [100, 0, 37, 90], (2, 3, 4), 0.8, 6, 3, False, 7, 0, 0),
    ([1500, 0, 700, 90], [100, 0, 37, 90], (3, 3, 3), 0.8, 32, 8, True, -1, 0, 1),
    ([1500, 0, 700, 90], [100, 0, 37, 90], (2, 2, 2), 0.8, 12, 4, True, 3, 1, 1),
    ([6000], [500], (4, 4, 4), 2.5, 128, 64, True, -1, 0, 0),                        # more than 64 channels per cell
    ([300, 300], [20, 20], (1, 1, 1), 0.5, 4, 4, True, -1, 1, 0),
]


@pytest.mark.parametrize("counts,mcounts,grid,dist,c_in,ceg,use_xyz,nsample,ntype,pooling", POOL_CASES)
def test_vector_pool_forward_backward(oracle, dev, counts, mcounts, grid, dist, c_in, ceg, use_xyz, nsample, ntype, pooling):
    xyz = ragged_clouds(counts, 8)
    new = centres_from(xyz, counts, mcounts, 6)
    new[3] += 200.0                                                                   # a centre that pools nothing
    rng = np.random.default_rng(2)
    feats = rng.normal(size=(xyz.shape[0], c_in)).astype(np.float32)
    ref = oracle.stack_vector_pool(xyz, counts, feats, new, mcounts, grid, dist, ceg, use_xyz, 3, nsample, ntype, pooling)
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    f = T(feats, dev).requires_grad_(True)
    nf, nl, mean, cnt = su.vector_pool_with_voxel_query_op(T(xyz, dev), xc, f, T(new, dev), nc, grid[0], grid[1], grid[2], dist, ceg,
                                                           use_xyz, 3, nsample, ntype, pooling)
    np.testing.assert_array_equal(cnt.cpu().numpy(), ref['point_cnt_of_grid'])
    np.testing.assert_array_equal(nf.detach().cpu().numpy(), ref['new_features'])
    np.testing.assert_array_equal(nl.cpu().numpy(), ref['new_local_xyz'])
    assert int(mean) == ref['num_mean_points_per_grid'] and mean.dtype == torch.int32 and not mean.is_cuda
    assert (ref['point_cnt_of_grid'][3] == 0).all() and ref['point_cnt_of_grid'].sum() > 0
    g = rng.normal(size=ref['new_features'].shape).astype(np.float32)
    nf.backward(T(g, dev))
    want = oracle.stack_vector_pool_grad(g, ref['point_cnt_of_grid'], ref['grouped_idxs'], xyz.shape[0], c_in)
    np.testing.assert_allclose(f.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


def test_vector_pool_wrapper_protocol_and_grouped_idxs(oracle, dev):
    """The pybind-level call: the return value is the number of entries wanted; with room for them grouped_idxs is the
    oracle's (centre order), without it nothing is written and the caller retries — upstream's loop runs unchanged."""
    counts, mcounts, grid = [900, 400], [40, 25], (3, 3, 3)
    xyz = ragged_clouds(counts, 12)
    new = centres_from(xyz, counts, mcounts, 1)
    feats = np.random.default_rng(4).normal(size=(1300, 8)).astype(np.float32)
    ref = oracle.stack_vector_pool(xyz, counts, feats, new, mcounts, grid, 1.0, 4, True, 2)
    M, G = 65, 27
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    mean, rounds = 2, 0
    while True:                                                                        # pointnet2_utils.py:396-414, verbatim protocol
        nf = torch.zeros((M, 4 * G), device=dev)
        nl = torch.zeros((M, 3 * G), device=dev)
        cnt = torch.zeros((M, G), dtype=torch.int32, device=dev)
        cap = mean * M
        grouped = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
        total = ext.vector_pool_wrapper(T(xyz, dev), xc, T(feats, dev), T(new, dev), nc, nf, nl, cnt, grouped, 3, 3, 3, 1.0, 1, cap,
                                        -1, 0, 0)
        mean = total // M + int(total % M > 0)
        rounds += 1
        if total <= cap:
            break
    assert rounds == 2 and total == len(ref['grouped_idxs'])
    np.testing.assert_array_equal(grouped[:total].cpu().numpy(), ref['grouped_idxs'])
    np.testing.assert_array_equal(cnt.cpu().numpy(), ref['point_cnt_of_grid'])
    norm = np.maximum(ref['point_cnt_of_grid'][:, :, None].astype(np.float32), np.float32(1e-6))
    np.testing.assert_array_equal((nf.cpu().numpy().reshape(M, G, 4) / norm).reshape(M, -1), ref['new_features'])


def test_vector_pool_rejects_bad_arguments(dev):
    xyz = torch.zeros((10, 3), device=dev)
    cnt = torch.tensor([10], dtype=torch.int32, device=dev)
    with pytest.raises(AssertionError):
        su.vector_pool_with_voxel_query_op(xyz, cnt, torch.zeros((10, 6), device=dev), xyz, cnt, 2, 2, 2, 1.0, 4, True)
    with pytest.raises(Exception, match="LDS|fit"):
        su.vector_pool_with_voxel_query_op(xyz, cnt, torch.zeros((10, 4096), device=dev), xyz, cnt, 2, 2, 2, 1.0, 4096, True)
    with pytest.raises(ValueError):
        ext.voxel_query_wrapper(1, 2, 2, 2, 4, 1.0, 1, 1, 1, xyz, xyz, torch.zeros((1, 4), dtype=torch.int32, device=dev),
                                torch.zeros((1, 3, 3, 3), dtype=torch.int32, device=dev), torch.zeros((1, 4), dtype=torch.int32, device=dev))


def test_vector_pool_large_cloud_properties(dev):
    """BASELINE-size cloud (no oracle: properties only): counts add up to the entries, every cell average lies inside
    the feature range, the average of a constant feature is the constant, and the gradient of sum(new_features) puts
    num_c_in / cell-count weight on each pooled point."""
    n, m = 65536, 4096
    xyz = T(ragged_clouds([n, n], 40), dev)
    xc = torch.tensor([n, n], dtype=torch.int32, device=dev)
    pick = torch.randperm(n, device=dev)[:m]
    new = torch.cat([xyz[:n][pick], xyz[n:][pick]]).contiguous()
    nc = torch.tensor([m, m], dtype=torch.int32, device=dev)
    f = torch.ones((2 * n, 16), device=dev, requires_grad=True)
    nf, nl, mean, cnt = su.vector_pool_with_voxel_query_op(xyz, xc, f, new, nc, 3, 3, 3, 1.6, 16, True, 50, -1, 0, 0)
    filled = (cnt > 0)
    assert torch.equal(nf.view(2 * m, 27, 16)[filled], torch.ones_like(nf.view(2 * m, 27, 16)[filled]))
    assert (nf.view(2 * m, 27, 16)[~filled] == 0).all() and (nl.abs() <= 1.6 + 1e-5).all()
    assert int(mean) == -(-int(cnt.sum()) // (2 * m))
    nf.sum().backward()
    # each pooled (point, centre) pair contributes 1 / count per channel; summed over a cell that is 1 per channel
    assert abs(float(f.grad.sum()) - float(filled.sum()) * 16) < 1e-3 * float(filled.sum()) * 16


# ---- the modules built on the operators (pointnet2_modules.py:160-470) ---------------------------------------------

def _module_inputs(dev, c=16):
    counts, mcounts = [1800, 1100], [70, 50]
    xyz = ragged_clouds(counts, 17)
    new = centres_from(xyz, counts, mcounts, 5)
    feats = np.random.default_rng(6).normal(size=(xyz.shape[0], c)).astype(np.float32)
    return counts, mcounts, xyz, new, feats


def test_local_interpolate_module_matches_oracle(oracle, dev):
    from pdm_ssd_amd.pointnet2_stack.pointnet2_modules import VectorPoolAggregationModule, VectorPoolLocalInterpolateModule
    counts, mcounts, xyz, new, feats = _module_inputs(dev)
    new[4] += 400.0                                                                   # no neighbours: its cells come out zero
    grid, R = (3, 3, 2), 0.9
    mod = VectorPoolLocalInterpolateModule(None, grid, R, -1, 1, use_xyz=True, neighbour_distance_multiplier=2.0).to(dev)
    centres_t = VectorPoolAggregationModule.get_dense_voxels_by_center(T(new, dev), R, grid)
    centres = centres_t.cpu().numpy()
    # lattice: x slowest, z fastest, cell centres at -R + (2i + 1) R / n
    want_axis = [(-R + (2 * np.arange(n) + 1) * R / n) for n in grid]
    off = np.stack(np.meshgrid(*want_axis, indexing="ij"), -1).reshape(-1, 3)
    np.testing.assert_allclose(centres, new[:, None, :] + off[None], atol=1e-5)
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    out = mod(T(xyz, dev), T(feats, dev), xc, T(new, dev), centres_t, nc).cpu().numpy()
    G = 18
    d, i, _ = oracle.stack_three_nn_for_vector_pool_by_two_step(xyz, counts, new, centres, mcounts, R, -1, 1, 1000, G, 2.0)
    d, i = d.reshape(-1, 3), i.reshape(-1, 3).copy()
    empty = i[:, 0] == -1
    i[empty] = 0
    with np.errstate(over="ignore", invalid="ignore"):
        rec = np.float32(1.0) / (d + np.float32(1e-8))
        w = rec / np.maximum(rec.sum(-1, keepdims=True), np.float32(1e-8))
    w[empty] = 0
    interp = oracle.stack_three_interpolate(feats, i, w.astype(np.float32))
    offsets = (centres.reshape(-1, 1, 3) - xyz[i]).reshape(-1, 9)
    want = np.concatenate([interp, offsets], 1)
    want[empty] = 0
    assert empty.reshape(-1, G)[4].all() and not empty.all()
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind", ["voxel_avg_pool", "voxel_random_choice", "local_interpolation"])
def test_vector_pool_aggregation_module(oracle, dev, kind):
    """The gather stage against the oracle, then the module's own layers (grouped 1x1 + BN + ReLU, post MLPs) against
    the same layers run on the CPU; the backward reaches the support features."""
    import copy
    from pdm_ssd_amd.pointnet2_stack.pointnet2_modules import VectorPoolAggregationModule
    counts, mcounts, xyz, new, feats = _module_inputs(dev, c=32)
    torch.manual_seed(0)
    mod = VectorPoolAggregationModule(32, (3, 3, 3), kind, 16, 8, (24, 20), 0.9, -1, 0, 2.0).eval()
    cpu = copy.deepcopy(mod)
    mod = mod.to(dev)
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    f = T(feats, dev).requires_grad_(True)
    key_xyz, out = mod(T(xyz, dev), xc, T(new, dev), nc, f)
    assert out.shape == (120, 20) and key_xyz.shape == (120, 3)
    red = feats.reshape(-1, 2, 16).sum(1)
    if kind == "local_interpolation":
        vec = mod.vector_pool_with_local_interpolate(T(xyz, dev), xc, T(red, dev), T(new, dev), nc).cpu()
        assert vec.shape == (120, 27 * 25)
    else:
        r = oracle.stack_vector_pool(xyz, counts, red, new, mcounts, (3, 3, 3), 0.9, 16, True, 20, -1, 0, 0 if kind == "voxel_avg_pool" else 1)
        vec = torch.from_numpy(np.concatenate([r['new_local_xyz'].reshape(120, 27, 3), r['new_features'].reshape(120, 27, 16)], -1).reshape(120, -1))
        got, cnt = mod.vector_pool_with_voxel_query(T(xyz, dev), xc, T(red, dev), T(new, dev), nc)
        np.testing.assert_allclose(got.cpu().numpy(), vec.numpy(), rtol=1e-6, atol=1e-6)   # red is summed on the GPU by torch
        np.testing.assert_array_equal(cnt.cpu().numpy(), r['point_cnt_of_grid'])
    with torch.no_grad():
        want = cpu.post_mlps(cpu.separate_local_aggregation_layer(vec.t()[None])).squeeze(0).t()
    np.testing.assert_allclose(out.detach().cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4)
    out.square().sum().backward()
    assert torch.isfinite(f.grad).all() and float(f.grad.abs().sum()) > 0


def test_vector_pool_msg_module_from_config(dev):
    from pdm_ssd_amd.pointnet2_stack.pointnet2_modules import build_local_aggregation_module
    cfg = {'NAME': 'VectorPoolAggregationModuleMSG', 'NUM_GROUPS': 2, 'LOCAL_AGGREGATION_TYPE': 'local_interpolation',
           'NUM_REDUCED_CHANNELS': 8, 'NUM_CHANNELS_OF_LOCAL_AGGREGATION': 16, 'MSG_POST_MLPS': [32],
           'GROUP_CFG_0': {'NUM_LOCAL_VOXEL': [2, 2, 2], 'MAX_NEIGHBOR_DISTANCE': 0.4, 'NEIGHBOR_NSAMPLE': -1, 'POST_MLPS': [32, 32]},
           'GROUP_CFG_1': {'NUM_LOCAL_VOXEL': [3, 3, 3], 'MAX_NEIGHBOR_DISTANCE': 0.8, 'NEIGHBOR_NSAMPLE': -1, 'POST_MLPS': [32, 32]}}
    layer, c_out = build_local_aggregation_module(16, cfg)
    assert c_out == 32 and {k.split('.')[0] for k in layer.state_dict()} == {'layer_0', 'layer_1', 'msg_post_mlps'}
    layer = layer.to(dev).eval()
    counts, mcounts, xyz, new, feats = _module_inputs(dev)
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    key_xyz, out = layer(xyz=T(xyz, dev), xyz_batch_cnt=xc, new_xyz=T(new, dev), new_xyz_batch_cnt=nc, features=T(feats, dev))
    assert out.shape == (120, 32) and torch.isfinite(out).all() and torch.equal(key_xyz, T(new, dev))
