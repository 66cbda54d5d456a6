"""pointnet2_stack operators (ragged batches): HIP kernels through the reference-shaped
python API against oracle/pointnet2_stack_oracle.c.  Indices bit-exact; copies bit-exact; interpolation 1e-6."""
import numpy as np
import pytest
import torch

from pdm_ssd_amd import synthetic
from pdm_ssd_amd.pointnet2_stack import pointnet2_utils as su

pytestmark = pytest.mark.gpu


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def ragged_clouds(counts, seed, kind="lidar"):
    gen = synthetic.lidar_like_clouds if kind == "lidar" else synthetic.uniform_clouds
    pts = [gen(1, max(n, 1), seed + i)[0, :n, :3] for i, n in enumerate(counts)]
    return np.ascontiguousarray(np.concatenate(pts).astype(np.float32))


CASES = [([1500, 700, 4096], [64, 33, 200]), ([16384, 3], [512, 2]), ([5, 0, 900], [5, 0, 100]), ([1024], [1024])]


@pytest.mark.parametrize("counts,npoints", CASES)
def test_stack_fps_index_exact(oracle, dev, counts, npoints):
    xyz = ragged_clouds(counts, 7)
    ref = oracle.stack_furthest_point_sample(xyz, counts, npoints)
    got = su.stack_farthest_point_sample(T(xyz, dev), T(np.array(counts, np.int32), dev), npoints)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_stack_fps_duplicates_follow_the_1024_thread_tree(oracle, dev):
    """Exact ties everywhere (every point duplicated many times): the pick order is the reference tree's."""
    rng = np.random.default_rng(0)
    base = rng.uniform(0, 5, (40, 3)).astype(np.float32)
    xyz = np.concatenate([np.tile(base, (30, 1)), np.tile(base[:7], (100, 1))])       # 1200 + 700 points
    counts = [1200, 700]
    ref = oracle.stack_furthest_point_sample(xyz, counts, [60, 20])
    got = su.stack_farthest_point_sample(T(xyz, dev), T(np.array(counts, np.int32), dev), [60, 20])
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    # int npoint = the same count for every sample; tensor npoint accepted too
    a = su.stack_farthest_point_sample(T(xyz, dev), T(np.array(counts, np.int32), dev), 16)
    b = su.stack_farthest_point_sample(T(xyz, dev), T(np.array(counts, np.int32), dev), torch.tensor([16, 16], device=dev))
    assert torch.equal(a, b) and a.numel() == 32


@pytest.mark.parametrize("radius,nsample", [(0.3, 16), (1.0, 32), (4.0, 7), (0.0, 4)])
def test_stack_ball_query_group_match_oracle(oracle, dev, radius, nsample):
    counts, mcounts = [1500, 0, 700, 90], [100, 0, 37, 90]
    xyz = ragged_clouds(counts, 11)
    rng = np.random.default_rng(1)
    starts = np.concatenate([[0], np.cumsum(counts)])
    new = np.concatenate([xyz[starts[b]:starts[b + 1]][rng.permutation(counts[b])[:m]] for b, m in enumerate(mcounts)])
    new[3] += 100.0                                                                    # an empty ball
    ridx, rmask = oracle.stack_ball_query(radius, nsample, xyz, counts, new, mcounts)
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    idx, mask = su.ball_query(radius, nsample, T(xyz, dev), xc, T(new, dev), nc)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(mask.cpu().numpy(), rmask)
    assert rmask[3]
    feats = rng.standard_normal((xyz.shape[0], 5)).astype(np.float32)
    ref = oracle.stack_grouping_operation(feats, counts, ridx, mcounts)
    f = T(feats, dev).requires_grad_(True)
    got = su.grouping_operation(f, xc, idx, nc)
    np.testing.assert_array_equal(got.detach().cpu().numpy(), ref)
    g = rng.standard_normal(ref.shape).astype(np.float32)
    got.backward(T(g, dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.stack_grouping_operation_grad(g, ridx, mcounts, counts, xyz.shape[0]),
                               rtol=1e-5, atol=1e-5)


def test_stack_query_and_group_module(oracle, dev):
    counts, mcounts = [800, 300], [50, 20]
    xyz = ragged_clouds(counts, 2)
    new = np.concatenate([xyz[:50], xyz[800:820]])
    new[7] += 50.0
    rng = np.random.default_rng(4)
    feats = rng.standard_normal((1100, 6)).astype(np.float32)
    qg = su.QueryAndGroup(0.8, 16, use_xyz=True)
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    out, idx = qg(T(xyz, dev), xc, T(new, dev), nc, T(feats, dev))
    ridx, rmask = oracle.stack_ball_query(0.8, 16, xyz, counts, new, mcounts)
    gx = oracle.stack_grouping_operation(xyz, counts, ridx, mcounts) - new[:, :, None]
    gf = oracle.stack_grouping_operation(feats, counts, ridx, mcounts)
    gx[rmask] = 0; gf[rmask] = 0
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(out.cpu().numpy(), np.concatenate([gx, gf], 1))
    assert out.shape == (70, 9, 16) and (out[7] == 0).all()


def test_stack_three_nn_interpolate_match_oracle(oracle, dev):
    ucounts, kcounts = [1000, 5, 400], [120, 2, 64]          # sample 1 has only two known points
    unknown, known = ragged_clouds(ucounts, 3), ragged_clouds(kcounts, 30)
    known[5] = known[6]                                       # an exact tie between two known points
    rd, ri = oracle.stack_three_nn(unknown, ucounts, known, kcounts)
    uc, kc = T(np.array(ucounts, np.int32), dev), T(np.array(kcounts, np.int32), dev)
    d, i = su.three_nn(T(unknown, dev), uc, T(known, dev), kc)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(d.cpu().numpy(), rd)
    assert np.isinf(rd[1000:1005, 2]).all()
    rng = np.random.default_rng(2)
    feats = rng.standard_normal((known.shape[0], 9)).astype(np.float32)
    w = rng.uniform(0, 1, ri.shape).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    f = T(feats, dev).requires_grad_(True)
    out = su.three_interpolate(f, i, T(w, dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), oracle.stack_three_interpolate(feats, ri, w), rtol=1e-6, atol=1e-6)
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(T(g, dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.stack_three_interpolate_grad(g, ri, w, known.shape[0]),
                               rtol=1e-4, atol=1e-5)


def test_stack_equals_batch_operators_on_equal_counts(dev):
    """The stacked operators on an equal-count batch reproduce the batch operators (shared arithmetic)."""
    from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as bu
    B, N, M = 4, 2048, 256
    xyz = torch.from_numpy(synthetic.lidar_like_clouds(B, N, 21)[:, :, :3].copy()).to(dev)
    flat = xyz.reshape(-1, 3).contiguous()
    cnt = torch.full((B,), N, dtype=torch.int32, device=dev)
    # N = 2048: both variants run 1024 logical threads, so even the tie order agrees
    fi = bu.furthest_point_sample(xyz, M)
    si = su.stack_farthest_point_sample(flat, cnt, M)
    assert torch.equal(si.view(B, M), fi + torch.arange(B, device=dev, dtype=torch.int32)[:, None] * N)
    new = torch.gather(xyz, 1, fi.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    mc = torch.full((B,), M, dtype=torch.int32, device=dev)
    bi = bu.ball_query(1.0, 32, xyz, new)
    si2, empty = su.ball_query(1.0, 32, flat, cnt, new.view(-1, 3).contiguous(), mc)
    assert torch.equal(si2.view(B, M, 32), bi) and not bool(empty.any())
    d, i = bu.three_nn(xyz, new)
    sd, si3 = su.three_nn(flat, cnt, new.view(-1, 3).contiguous(), mc)
    assert torch.equal(sd.view(B, N, 3), d)
    assert torch.equal(si3.view(B, N, 3), i + torch.arange(B, device=dev, dtype=torch.int32)[:, None, None] * M)


def test_stack_wrappers_reject_misuse(dev):
    from pdm_ssd_amd.pointnet2_stack import pointnet2_stack_hip as sh
    xyz = torch.rand(10, 3, device=dev)
    with pytest.raises(TypeError):
        sh.ball_query_wrapper(1, 10, 1.0, 4, xyz, torch.tensor([10], device=dev), xyz,
                              torch.tensor([10], device=dev, dtype=torch.int32), torch.zeros(10, 4, dtype=torch.int32, device=dev))
    with pytest.raises(ValueError):
        su.stack_farthest_point_sample(xyz, torch.tensor([4, 4], dtype=torch.int32, device=dev), 2)   # counts != rows
    with pytest.raises(ValueError):
        su.stack_farthest_point_sample(xyz, torch.tensor([10, 0], dtype=torch.int32, device=dev), [2, 1])


def _randomize_bn(module, seed):
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


@pytest.mark.parametrize("cin", [0, 5, 32])
def test_stack_sa_module_fused_equals_unfused(dev, cin):
    """StackSAModuleMSG in eval/no-grad (fused MFMA kernels on the stacked batch as one sample, empty balls patched)
    against its own torch path; 32 input channels also exercises the hoisted first layer."""
    from pdm_ssd_amd.pointnet2_stack import pointnet2_modules as sm
    torch.manual_seed(cin)
    counts, mcounts = [900, 300, 1200], [40, 25, 64]
    xyz = ragged_clouds(counts, 5)
    starts = np.concatenate([[0], np.cumsum(counts)])
    new = np.concatenate([xyz[starts[b]:starts[b] + m] for b, m in enumerate(mcounts)])
    new[10] += 200.0                                                  # an empty ball in sample 0
    feats = None if cin == 0 else torch.randn(sum(counts), cin, device=dev)
    sa = sm.StackSAModuleMSG(radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[cin, 16, 32], [cin, 32, 48]]).to(dev).eval()
    _randomize_bn(sa, 3)
    xc, nc = T(np.array(counts, np.int32), dev), T(np.array(mcounts, np.int32), dev)
    with torch.no_grad():
        _, a = sa(T(xyz, dev), xc, T(new, dev), nc, feats)
        assert '_pdm_fused_cache' in sa.__dict__
        sa.use_fused = False
        _, b = sa(T(xyz, dev), xc, T(new, dev), nc, feats)
    assert a.shape == b.shape == (129, 80)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    sa.train()                                                        # autograd path
    f = None if feats is None else feats.clone().requires_grad_(True)
    _, c = sa(T(xyz, dev), xc, T(new, dev), nc, f)
    c.sum().backward()
    assert all(p.grad is not None for p in sa.parameters()) and (f is None or f.grad is not None)


def test_stack_fp_module_fused_equals_unfused(dev):
    from pdm_ssd_amd.pointnet2_stack import pointnet2_modules as sm
    torch.manual_seed(1)
    ucounts, kcounts = [700, 260], [90, 40]
    unknown, known = ragged_clouds(ucounts, 8), ragged_clouds(kcounts, 80)
    fp = sm.StackPointnetFPModule(mlp=[24 + 6, 32, 16]).to(dev).eval()
    _randomize_bn(fp, 4)
    uf, kf = torch.randn(960, 6, device=dev), torch.randn(130, 24, device=dev)
    uc, kc = T(np.array(ucounts, np.int32), dev), T(np.array(kcounts, np.int32), dev)
    with torch.no_grad():
        a = fp(T(unknown, dev), uc, T(known, dev), kc, uf, kf)
        fp.use_fused = False
        b = fp(T(unknown, dev), uc, T(known, dev), kc, uf, kf)
    assert a.shape == (960, 16)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)


def test_build_local_aggregation_module_contract(dev):
    from pdm_ssd_amd.pointnet2_stack import pointnet2_modules as sm
    cfg = {'NAME': 'StackSAModuleMSG', 'MLPS': [[16, 16], [16, 32]], 'POOL_RADIUS': [0.4, 0.8], 'NSAMPLE': [16, 16]}
    layer, cout = sm.build_local_aggregation_module(8, cfg)
    assert cout == 48 and cfg['MLPS'][0][0] == 11                    # input channels prepended, +3 for xyz, in place
    assert sorted(layer.state_dict())[0] == 'mlps.0.0.weight' and layer.mlps[0][0].weight.shape == (16, 11, 1, 1)
    with pytest.raises(NotImplementedError):
        sm.build_local_aggregation_module(8, {'NAME': 'SomethingElse'})                 # the reference's own fall-through (ref :26)
