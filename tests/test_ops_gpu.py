"""Parity of the HIP operators (through the C ABI) against the CPU oracle.

Bar: bit-exact for indices (FPS, ball_query, three_nn) and for pure copies (gather, group_points);
<= 1e-4 for fp32 features whose summation order is free (atomic backward passes); three_interpolate
is pinned to the oracle's rounding sequence and compared bit-exactly too.
"""
import numpy as np
import pytest
import torch

from pdm_ssd_amd import synthetic
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu

pytestmark = pytest.mark.gpu

SA_SCALES = [(0.1, 16), (0.5, 32), (1.0, 32), (2.0, 16), (4.0, 32)]


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def clouds(kind, B, N, seed=1234):
    f = synthetic.uniform_clouds if kind == "uniform" else synthetic.lidar_like_clouds
    return f(B, N, seed)[:, :, :3].copy()


# ------------------------------------------------------------------ FPS

@pytest.mark.parametrize("kind", ["uniform", "lidar"])
@pytest.mark.parametrize("N,m", [(1024, 256), (4096, 1024), (256, 64), (1000, 100), (3000, 64), (64, 64), (37, 9), (1, 1)])
def test_fps_index_exact(oracle, dev, kind, N, m):
    xyz = clouds(kind, 2, N, seed=7)
    ref, ref_temp = oracle.furthest_point_sample(xyz, m, return_temp=True)
    got = pu.furthest_point_sample(T(xyz, dev), m)
    assert got.dtype == torch.int32 and tuple(got.shape) == (2, m)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_fps_16384_full_size(oracle, dev):
    xyz = clouds("lidar", 2, 16384, seed=11)
    ref = oracle.furthest_point_sample(xyz, 4096)
    got = pu.furthest_point_sample(T(xyz, dev), 4096)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_fps_ties_duplicated_points(oracle, dev):
    """Padded clouds repeat points (data_processor.py:206-210): equal maxima are decided by the
    reference's tree order, not by the smallest index."""
    rng = np.random.default_rng(3)
    base = rng.uniform(-20, 20, (2, 1024, 3)).astype(np.float32)
    xyz = np.concatenate([base, base[:, ::-1]], axis=1)  # every point twice, N = 2048
    ref = oracle.furthest_point_sample(xyz, 512)
    got = pu.furthest_point_sample(T(xyz, dev), 512)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    # lattice: many exactly equal distances
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(8), indexing="ij"), -1).reshape(1, -1, 3)
    g = g.astype(np.float32)
    ref = oracle.furthest_point_sample(g, 300)
    got = pu.furthest_point_sample(T(g, dev), 300)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_fps_streaming_variant_large_n(oracle, dev):
    xyz = clouds("uniform", 1, 20000, seed=5)
    ref = oracle.furthest_point_sample(xyz, 128)
    got = pu.furthest_point_sample(T(xyz, dev), 128)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_fps_temp_holds_final_min_distances(oracle, dev):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    xyz = clouds("uniform", 2, 2048, seed=9)
    ref, ref_temp = oracle.furthest_point_sample(xyz, 64, return_temp=True)
    x = T(xyz, dev)
    temp = torch.full((2, 2048), 1e10, device=dev)
    idx = torch.empty((2, 64), dtype=torch.int32, device=dev)
    ext.farthest_point_sampling_wrapper(2, 2048, 64, x, temp, idx)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(temp.cpu().numpy(), ref_temp)


# ------------------------------------------------------------------ ball query

@pytest.mark.parametrize("kind", ["uniform", "lidar"])
@pytest.mark.parametrize("N,M", [(4096, 1024), (1024, 256), (1000, 77), (256, 64)])
@pytest.mark.parametrize("radius,ns", SA_SCALES)
def test_ball_query_index_exact(oracle, dev, kind, N, M, radius, ns):
    xyz = clouds(kind, 2, N, seed=21)
    fidx = oracle.furthest_point_sample(xyz, M)
    new_xyz = np.take_along_axis(xyz, fidx[:, :, None].astype(np.int64), 1)
    ref = oracle.ball_query(radius, ns, xyz, new_xyz)
    got = pu.ball_query(radius, ns, T(xyz, dev), T(new_xyz, dev))
    assert got.dtype == torch.int32
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_ball_query_empty_balls_stay_zero(oracle, dev):
    xyz = clouds("uniform", 2, 512, seed=1)
    new_xyz = xyz[:, :32].copy()
    new_xyz[:, ::2] += 1000.0  # far away: no neighbour
    ref = oracle.ball_query(0.5, 8, xyz, new_xyz)
    got = pu.ball_query(0.5, 8, T(xyz, dev), T(new_xyz, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    assert (got[:, ::2] == 0).all()


def test_ball_query_odd_nsample_and_16384(oracle, dev):
    xyz = clouds("lidar", 1, 16384, seed=2)
    fidx = oracle.furthest_point_sample(xyz, 512)
    new_xyz = np.take_along_axis(xyz, fidx[:, :, None].astype(np.int64), 1)
    for radius, ns in [(0.5, 32), (0.3, 7), (2.0, 100)]:
        ref = oracle.ball_query(radius, ns, xyz, new_xyz)
        got = pu.ball_query(radius, ns, T(xyz, dev), T(new_xyz, dev))
        np.testing.assert_array_equal(got.cpu().numpy(), ref)


# ------------------------------------------------------------------ gather / group

def test_gather_and_grad(oracle, dev):
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((3, 5, 700)).astype(np.float32)
    idx = rng.integers(0, 700, (3, 123)).astype(np.int32)
    f = T(feat, dev).requires_grad_(True)
    out = pu.gather_operation(f, T(idx, dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.gather_operation(feat, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(T(go, dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.gather_operation_grad(go, idx, 700), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("C,N,M,ns", [(1, 1024, 256, 16), (96, 4096, 256, 32), (7, 300, 33, 5), (3, 128, 16, 4), (17, 64, 1, 1)])
def test_group_points_and_grad(oracle, dev, C, N, M, ns):
    rng = np.random.default_rng(C * 1000 + ns)
    feat = rng.standard_normal((2, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (2, M, ns)).astype(np.int32)
    f = T(feat, dev).requires_grad_(True)
    out = pu.grouping_operation(f, T(idx, dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.grouping_operation(feat, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(T(go, dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.grouping_operation_grad(go, idx, N), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("C", [0, 1, 29])
@pytest.mark.parametrize("ns", [16, 6])
def test_query_and_group_fused_matches_unfused_and_oracle(oracle, dev, C, ns):
    xyz = clouds("lidar", 2, 2048, seed=4)
    fidx = oracle.furthest_point_sample(xyz, 200)
    new_xyz = np.take_along_axis(xyz, fidx[:, :, None].astype(np.int64), 1)
    rng = np.random.default_rng(8)
    feat = rng.standard_normal((2, C, 2048)).astype(np.float32) if C else None
    ref, ref_idx = oracle.query_and_group(1.0, ns, xyz, new_xyz, feat)
    x, nx = T(xyz, dev), T(new_xyz, dev)
    f = T(feat, dev).requires_grad_(True) if C else None
    fused = pu.QueryAndGroup(1.0, ns, use_xyz=True, fused=True)(x, nx, f)
    np.testing.assert_array_equal(fused.detach().cpu().numpy(), ref)
    f2 = T(feat, dev).requires_grad_(True) if C else None
    unfused = pu.QueryAndGroup(1.0, ns, use_xyz=True, fused=False)(x, nx, f2)
    np.testing.assert_array_equal(unfused.detach().cpu().numpy(), ref)
    if C:
        go = T(rng.standard_normal(ref.shape).astype(np.float32), dev)
        fused.backward(go)
        unfused.backward(go)
        np.testing.assert_allclose(f.grad.cpu().numpy(), f2.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(f.grad.cpu().numpy(),
                                   oracle.grouping_operation_grad(go[:, 3:].cpu().numpy(), ref_idx, 2048),
                                   rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ three_nn / interpolate

@pytest.mark.parametrize("kind", ["uniform", "lidar"])
@pytest.mark.parametrize("n,m", [(4096, 1024), (1024, 256), (333, 50), (64, 3), (16, 2)])
def test_three_nn_index_exact(oracle, dev, kind, n, m):
    unknown = clouds(kind, 2, n, seed=31)
    known = clouds(kind, 2, m, seed=32) if m < 3 or n == 333 else unknown[:, :m].copy()
    ref_d, ref_i = oracle.three_nn(unknown, known)
    d, i = pu.three_nn(T(unknown, dev), T(known, dev))
    np.testing.assert_array_equal(i.cpu().numpy(), ref_i)
    np.testing.assert_array_equal(d.cpu().numpy(), ref_d)  # sqrt of identical fp32 squared distances


def test_three_nn_exact_ties_prefer_lower_index(oracle, dev):
    known = np.zeros((1, 8, 3), dtype=np.float32)
    known[0, :, 0] = [1, -1, 1, -1, 2, 2, -2, 3]  # distances from origin: 1,1,1,1,4,4,4,9
    unknown = np.zeros((1, 4, 3), dtype=np.float32)
    ref_d, ref_i = oracle.three_nn(unknown, known)
    np.testing.assert_array_equal(ref_i[0, 0], [0, 1, 2])
    d, i = pu.three_nn(T(unknown, dev), T(known, dev))
    np.testing.assert_array_equal(i.cpu().numpy(), ref_i)


@pytest.mark.parametrize("C,m,n", [(1024, 64, 256), (256, 1024, 4096), (5, 33, 77), (9, 4096, 9000), (3, 20000, 500)])
def test_three_interpolate_and_grad(oracle, dev, C, m, n):
    rng = np.random.default_rng(C)
    feat = rng.standard_normal((2, C, m)).astype(np.float32)
    idx = rng.integers(0, m, (2, n, 3)).astype(np.int32)
    w = rng.uniform(0, 1, (2, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    f = T(feat, dev).requires_grad_(True)
    out = pu.three_interpolate(f, T(idx, dev), T(w, dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.three_interpolate(feat, idx, w))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(T(go, dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.three_interpolate_grad(go, idx, w, m), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ boundary behaviour

def test_wrapper_rejects_bad_tensors(dev):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    xyz = torch.zeros((1, 8, 3), device=dev)
    idx = torch.zeros((1, 2, 4), dtype=torch.int32, device=dev)
    with pytest.raises(ValueError):
        ext.ball_query_wrapper(1, 8, 2, 1.0, 4, xyz[:, :2].cpu(), xyz, idx)  # not on the GPU
    with pytest.raises(ValueError):
        ext.ball_query_wrapper(1, 8, 2, 1.0, 4, xyz.transpose(1, 2)[:, :2], xyz, idx)  # not contiguous
    with pytest.raises(TypeError):
        ext.ball_query_wrapper(1, 8, 2, 1.0, 4, xyz[:, :2].contiguous(), xyz, idx.long())
    with pytest.raises(ValueError):
        ext.ball_query_wrapper(1, 8, 3, 1.0, 4, xyz[:, :2].contiguous(), xyz, idx)  # idx too small for m=3


def test_native_argument_errors_raise(dev):
    from pdm_ssd_amd import _native
    with pytest.raises(_native.NativeLibraryError):
        _native.call("pdm_ball_query", 0, -1, 8, 2, 1.0, 4, 0, 0, 0)
    with pytest.raises(_native.NativeLibraryError):
        _native.call("pdm_ball_query", 0, 1, 8, 2, 1.0, 4, 0, 0, 0)  # null pointers


def test_ops_run_on_non_default_stream(oracle, dev):
    xyz = clouds("uniform", 2, 1024, seed=77)
    ref = oracle.furthest_point_sample(xyz, 128)
    s = torch.cuda.Stream()
    x = T(xyz, dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        got = pu.furthest_point_sample(x, 128)
    s.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


# ------------------------------------------------------------------ grid-accelerated ball query

@pytest.fixture
def force_grid(monkeypatch):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    monkeypatch.setattr(ext, "GRID_MIN_N", 1)
    return ext


@pytest.fixture
def force_scan(monkeypatch):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    monkeypatch.setattr(ext, "GRID_MIN_N", 1 << 30)
    return ext


@pytest.mark.parametrize("kind", ["uniform", "lidar"])
@pytest.mark.parametrize("N,M", [(16384, 1024), (4096, 1024), (1000, 77), (70, 70)])
@pytest.mark.parametrize("radius,ns", [(0.1, 16), (0.5, 32), (2.0, 16), (4.0, 32), (200.0, 48), (0.0, 8)])
def test_ball_query_grid_index_exact(oracle, dev, force_grid, kind, N, M, radius, ns):
    xyz = clouds(kind, 2, N, seed=41)
    fidx = oracle.furthest_point_sample(xyz, M)
    new_xyz = np.take_along_axis(xyz, fidx[:, :, None].astype(np.int64), 1)
    new_xyz[:, ::7] += np.float32(radius * 0.6)  # centres that are not cloud points
    ref = oracle.ball_query(radius, ns, xyz, new_xyz)
    got = pu.ball_query(radius, ns, T(xyz, dev), T(new_xyz, dev))
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_ball_query_grid_degenerate_clouds(oracle, dev, force_grid):
    rng = np.random.default_rng(5)
    # all points identical; points on a line; far outliers stretching the bounding box; duplicates
    same = np.ones((1, 300, 3), dtype=np.float32) * 3.25
    line = np.zeros((1, 300, 3), dtype=np.float32); line[0, :, 0] = np.linspace(0, 30, 300)
    outl = rng.uniform(0, 5, (1, 300, 3)).astype(np.float32); outl[0, 7] = [1e6, -1e6, 1e5]
    dup = np.repeat(rng.uniform(0, 5, (1, 150, 3)).astype(np.float32), 2, axis=1)
    for xyz in (same, line, outl, dup):
        new_xyz = np.ascontiguousarray(xyz[:, ::3])
        for radius, ns in [(0.5, 16), (3.0, 64), (1e7, 32)]:
            ref = oracle.ball_query(radius, ns, xyz, new_xyz)
            got = pu.ball_query(radius, ns, T(xyz, dev), T(new_xyz, dev))
            np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_ball_query_grid_equals_scan_at_full_size(dev, monkeypatch):
    """Size-independent property at BASELINE's full size: both kernels give identical indices."""
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    xyz = T(clouds("lidar", 8, 16384, seed=51), dev)
    new_xyz = xyz[:, :4096].contiguous()
    for radius, ns in [(0.1, 16), (0.5, 32), (1.0, 32)]:
        monkeypatch.setattr(ext, "GRID_MIN_N", 1)
        a = pu.ball_query(radius, ns, xyz, new_xyz)
        monkeypatch.setattr(ext, "GRID_MIN_N", 1 << 30)
        b = pu.ball_query(radius, ns, xyz, new_xyz)
        assert torch.equal(a, b)


def test_ball_query_scan_path_still_exact(oracle, dev, force_scan):
    xyz = clouds("lidar", 2, 4096, seed=61)
    new_xyz = np.ascontiguousarray(xyz[:, :512])
    ref = oracle.ball_query(0.8, 32, xyz, new_xyz)
    got = pu.ball_query(0.8, 32, T(xyz, dev), T(new_xyz, dev))
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


# ------------------------------------------------------------------ grid-accelerated three_nn

@pytest.mark.parametrize("kind", ["uniform", "lidar"])
@pytest.mark.parametrize("n,m", [(16384, 4096), (4096, 1024), (1000, 600), (300, 3), (50, 1), (64, 2)])
def test_three_nn_grid_index_exact(oracle, dev, monkeypatch, kind, n, m):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    monkeypatch.setattr(ext, "NN_GRID_MIN_M", 1)
    unknown = clouds(kind, 2, n, seed=71)
    known = np.ascontiguousarray(unknown[:, :m]) if m >= 600 else clouds(kind, 2, m, seed=72)
    ref_d2, ref_i = oracle.three_nn_dist2(unknown, known)
    d, i = pu.three_nn(T(unknown, dev), T(known, dev))
    np.testing.assert_array_equal(i.cpu().numpy(), ref_i)
    np.testing.assert_array_equal(d.cpu().numpy(), np.sqrt(ref_d2))


def test_three_nn_grid_ties_duplicates_and_outliers(oracle, dev, monkeypatch):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    monkeypatch.setattr(ext, "NN_GRID_MIN_M", 1)
    rng = np.random.default_rng(9)
    lattice = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(4), indexing="ij"), -1).reshape(1, -1, 3)
    lattice = lattice.astype(np.float32)                       # many exactly equal distances
    dup = np.repeat(rng.uniform(0, 9, (1, 200, 3)).astype(np.float32), 3, axis=1)   # triplicated points
    same = np.full((1, 100, 3), 2.5, dtype=np.float32)          # zero extent
    outl = rng.uniform(0, 5, (1, 400, 3)).astype(np.float32); outl[0, 11] = [1e6, 1e6, -1e6]
    for known in (lattice, dup, same, outl):
        unknown = np.concatenate([known[:, ::2] + np.float32(0.5), rng.uniform(-3, 14, (1, 333, 3)).astype(np.float32)], 1)
        unknown = np.ascontiguousarray(unknown)
        ref_d2, ref_i = oracle.three_nn_dist2(unknown, known)
        d, i = pu.three_nn(T(unknown, dev), T(known, dev))
        np.testing.assert_array_equal(i.cpu().numpy(), ref_i)
        np.testing.assert_array_equal(d.cpu().numpy(), np.sqrt(ref_d2))


def test_three_nn_grid_equals_scan_at_full_size(dev, monkeypatch):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    xyz = T(clouds("lidar", 8, 16384, seed=81), dev)
    known = xyz[:, :4096].contiguous()
    monkeypatch.setattr(ext, "NN_GRID_MIN_M", 1)
    d1, i1 = pu.three_nn(xyz, known)
    monkeypatch.setattr(ext, "NN_GRID_MIN_M", 1 << 30)
    d2, i2 = pu.three_nn(xyz, known)
    assert torch.equal(i1, i2) and torch.equal(d1, d2)


# ------------------------------------------------------------------ pruned FPS (8192 < N <= 16384)

@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_fps_large_variants_index_exact(oracle, dev, variant):
    """All three kernels for 8192 < N <= 16384 (pruned / 512x32 / 1024x16) give the oracle's indices,
    including clouds with duplicated points, a lattice (many equal distances) and a ragged N."""
    from pdm_ssd_amd import _native
    rng = np.random.default_rng(17)
    base = clouds("lidar", 1, 8192, seed=91)
    dup = np.concatenate([base, base[:, ::-1]], axis=1)                      # every point twice
    lattice = np.stack(np.meshgrid(np.arange(32), np.arange(32), np.arange(16), indexing="ij"), -1)
    lattice = lattice.reshape(1, -1, 3).astype(np.float32)                   # 16384 grid points
    ragged = clouds("uniform", 2, 10007, seed=92)
    same = np.full((1, 9000, 3), 1.5, dtype=np.float32)                      # zero extent
    old = _native.lib().pdm_tune_fps_variant(variant)
    try:
        for xyz, m in ((dup, 700), (lattice, 600), (ragged, 500), (same, 40), (clouds("uniform", 2, 16384, seed=93), 2048)):
            ref, ref_temp = oracle.furthest_point_sample(xyz, m, return_temp=True)
            x = T(xyz, dev)
            got = pu.furthest_point_sample(x, m)
            np.testing.assert_array_equal(got.cpu().numpy(), ref)
    finally:
        _native.lib().pdm_tune_fps_variant(old)


def test_fps_pruned_leaves_final_min_distances_in_temp(oracle, dev):
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    xyz = clouds("lidar", 2, 12000, seed=94)
    ref, ref_temp = oracle.furthest_point_sample(xyz, 300, return_temp=True)
    x = T(xyz, dev)
    temp = torch.full((2, 12000), 1e10, device=dev)
    idx = torch.empty((2, 300), dtype=torch.int32, device=dev)
    ext.farthest_point_sampling_wrapper(2, 12000, 300, x, temp, idx)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(temp.cpu().numpy(), ref_temp)


# ------------------------------------------------------------------ multi-workgroup FPS (N > 16384)

@pytest.mark.parametrize("N,m", [(20000, 300), (65536, 400), (40001, 257)])
def test_fps_multi_workgroup_index_exact(oracle, dev, N, m):
    rng = np.random.default_rng(N)
    xyz = clouds("lidar", 2, N, seed=101)
    xyz[1, N // 2:] = xyz[1, :N - N // 2]      # second cloud: duplicates that straddle the workgroup split
    ref, ref_temp = oracle.furthest_point_sample(xyz, m, return_temp=True)
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    x = T(xyz, dev)
    temp = torch.full((2, N), 1e10, device=dev)
    idx = torch.empty((2, m), dtype=torch.int32, device=dev)
    ext.farthest_point_sampling_wrapper(2, N, m, x, temp, idx)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(temp.cpu().numpy(), ref_temp)


def test_fps_multi_workgroup_equals_stream_kernel(dev):
    """The API-exact entry point (no workspace) takes the single-workgroup streaming kernel; same answer."""
    from pdm_ssd_amd import _native
    xyz = T(clouds("uniform", 1, 30000, seed=103), dev)
    a = pu.furthest_point_sample(xyz, 200)
    temp = torch.full((1, 30000), 1e10, device=dev)
    idx = torch.empty((1, 200), dtype=torch.int32, device=dev)
    _native.call("pdm_furthest_point_sampling", torch.cuda.current_stream().cuda_stream, 1, 30000, 200,
                 xyz.data_ptr(), temp.data_ptr(), idx.data_ptr())
    assert torch.equal(a, idx)


def test_three_nn_weights_matches_reference_glue(dev):
    """pdm_three_nn_weights == the reference's python glue (sqrt, 1/(d+1e-8), normalise), same fp32 operations."""
    from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
    torch.manual_seed(0)
    unknown = (torch.rand(3, 1000, 3, device=dev) * 20).contiguous()
    known = unknown[:, ::7].contiguous()                       # some unknown points coincide with a known one (d = 0)
    dist, idx = pu.three_nn(unknown, known)
    dist_recip = 1.0 / (dist + 1e-8)
    ref = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
    idx2, w = pu.three_nn_weights(unknown, known)
    assert torch.equal(idx, idx2)
    torch.testing.assert_close(w, ref, rtol=1e-6, atol=1e-7)
    assert torch.isfinite(w).all() and torch.allclose(w.sum(-1), torch.ones_like(w[..., 0]), atol=1e-6)
    e_idx, e_w = pu.three_nn_weights(unknown[:, :0].contiguous(), known)
    assert e_idx.shape == (3, 0, 3) and e_w.shape == (3, 0, 3)


def test_copy_many_one_launch(dev):
    from pdm_ssd_amd import _native
    torch.manual_seed(1)
    shapes = [(3, 5), (1,), (0, 4), (1000, 3), (7, 11, 13), (2, 4096, 32)] * 10     # 60 buffers: two launches of <= 48
    src = [torch.randn(s, device=dev) if i % 2 else torch.randint(0, 1 << 30, s, device=dev, dtype=torch.int32)
           for i, s in enumerate(shapes)]
    src[3] = torch.randn(1001, 3, device=dev)[1:]                                    # 12-byte offset: unaligned path
    dst = [torch.full_like(t, 7) for t in src]
    _native.copy_many(dst, src)
    for d, s_ in zip(dst, src):
        assert torch.equal(d, s_)
    _native.copy_many([], [])


@pytest.mark.parametrize("n,m,cuts", [(16384, 1024, [1, 300, 1024]), (5000, 700, [1, 2, 350, 351, 700]), (2048, 64, [1, 64]),
                                      (40000, 300, [1, 120, 300])])   # last: three cooperating workgroups per cloud
def test_fps_resumable_segments_match_one_call(dev, n, m, cuts):
    """pdm_furthest_point_sampling_jobs: the segments of one batch, run in order (each beside a segment of ANOTHER
    batch in the same launch), give exactly the indices and final min-distances of the one-call operator."""
    cl_a = torch.from_numpy(np.ascontiguousarray(synthetic.lidar_like_clouds(3, n, 77)[:, :, :3])).to(dev)
    cl_b = torch.from_numpy(np.ascontiguousarray(synthetic.uniform_clouds(3, n, 78)[:, :, :3])).to(dev)
    want_a, want_b = pu.furthest_point_sample(cl_a, m), pu.furthest_point_sample(cl_b, m)
    st = {k: (torch.full((3, n), 1e10, device=dev), torch.full((3, m), -1, dtype=torch.int32, device=dev)) for k in "ab"}
    segs = list(zip(cuts[:-1], cuts[1:]))
    # batch b runs one segment behind batch a, so most launches hold two jobs at different stages
    for step in range(len(segs) + 1):
        jobs = []
        if step < len(segs):
            jobs.append((cl_a, st["a"][0], st["a"][1], segs[step][0], segs[step][1]))
        if step >= 1:
            jobs.append((cl_b, st["b"][0], st["b"][1], segs[step - 1][0], segs[step - 1][1]))
        pu.fps_segments(jobs, m)
    assert torch.equal(st["a"][1], want_a) and torch.equal(st["b"][1], want_b)
    ref_temp = torch.full((3, n), 1e10, device=dev)
    ref_idx = torch.empty((3, m), dtype=torch.int32, device=dev)
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as hipops
    hipops.farthest_point_sampling_wrapper(3, n, m, cl_a, ref_temp, ref_idx)
    assert torch.equal(st["a"][0], ref_temp)


def test_copy_many_device_side_lengths(dev):
    """pdm_copy_many_dyn: buffers with a live-row count on the device are copied up to that count only; the others
    whole; counts beyond the buffer or below zero are clamped."""
    from pdm_ssd_amd import _native
    src = [torch.arange(1000, dtype=torch.int32, device=dev).view(500, 2) + 1, torch.rand(77, device=dev),
           torch.arange(64, dtype=torch.int32, device=dev).view(32, 2) + 5, torch.rand(9, 3, device=dev)]
    dst = [torch.zeros_like(t) for t in src]
    counts = torch.tensor([0, 0, 0, 0, 0, 0, 123, 0, 10 ** 6, -4], dtype=torch.int32, device=dev)
    live = [(counts[6:7], 8), None, (counts[8:9], 8), (counts[9:10], 12)]
    _native.copy_many(dst, src, live)
    assert torch.equal(dst[0][:123], src[0][:123]) and (dst[0][123:] == 0).all()
    assert torch.equal(dst[1], src[1])
    assert torch.equal(dst[2], src[2])            # count larger than the buffer: all of it
    assert (dst[3] == 0).all()                    # negative count: nothing


def test_mark_time_orders_stream_work(dev):
    from pdm_ssd_amd import _native
    marks = torch.zeros(2, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _native.call("pdm_mark_time", s, marks.data_ptr())
    torch.cuda._sleep(200000)
    _native.call("pdm_mark_time", s, marks.data_ptr() + 8)
    torch.cuda.synchronize()
    t0, t1 = marks.cpu().tolist()
    assert t1 > t0 > 0


@pytest.mark.parametrize("c,n,m", [(5, 16384, 4096), (19, 5000, 1300), (8, 4096, 1024), (3, 300, 7), (33, 64, 64)])
def test_three_interpolate_grad_csr_form(oracle, dev, c, n, m):
    """The backward through the inverted (CSR) scatter — what three_interpolate_grad_wrapper runs — against the oracle,
    including the reference's accumulate-into-grad_points behaviour (the caller zero-fills, interpolate_gpu.cu:127-149)."""
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as hipops
    rng = np.random.default_rng(c * n + m)
    B = 3
    go = rng.standard_normal((B, c, n)).astype(np.float32)
    idx = rng.integers(0, m, size=(B, n, 3)).astype(np.int32)
    idx[0, : n // 2] = 0                                   # a heavily shared known point (long CSR list)
    w = rng.random((B, n, 3)).astype(np.float32)
    want = oracle.three_interpolate_grad(go, idx, w, m)
    gp = torch.full((B, c, m), 0.5, device=dev)
    hipops.three_interpolate_grad_wrapper(B, c, n, m, T(go, dev), T(idx, dev), T(w, dev), gp)
    scale = float(np.abs(want).max())
    np.testing.assert_allclose(gp.cpu().numpy() - 0.5, want, rtol=1e-4, atol=1e-5 * max(scale, 1.0) + 1e-4)


@pytest.mark.parametrize("c,n,m,ns", [(7, 4096, 1024, 32), (96, 2000, 1024, 16), (3, 16384, 512, 16), (5, 300, 64, 32), (4, 16384, 4096, 16)])
def test_group_points_grad_csr_form(oracle, dev, c, n, m, ns):
    """group_points backward through the inverted (CSR) scatter (rows of m*ns <= 32768 floats) and through the plain
    entry point beyond that (last case), against the oracle; grad_points is accumulated into, as the reference does."""
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as hipops
    rng = np.random.default_rng(c + n + m)
    B = 2
    go = rng.standard_normal((B, c, m, ns)).astype(np.float32)
    idx = rng.integers(0, n, size=(B, m, ns)).astype(np.int32)
    idx[1, :, : ns // 2] = 3                                # one source point in half of all slots
    want = oracle.grouping_operation_grad(go, idx, n)
    gp = torch.full((B, c, n), -1.0, device=dev)
    hipops.group_points_grad_wrapper(B, c, n, m, ns, T(go, dev), T(idx, dev), gp)
    scale = float(np.abs(want).max())
    np.testing.assert_allclose(gp.cpu().numpy() + 1.0, want, rtol=1e-4, atol=1e-5 * max(scale, 1.0) + 1e-4)


def test_shared_search_grid_scope_is_safe(oracle, dev):
    """Inside `with pu.shared_search_grids():` one grid per point set serves both MSG radii and the three_nn over it
    (pointnet2_batch_hip.GRID_CACHE); outside a scope nothing is kept, so a point set rewritten through a raw pointer
    (pdm_copy_many, a hipGraph replay — neither bumps the tensor's `_version`) is searched on a grid of its CURRENT
    coordinates; inside a scope an in-place torch write and a new tensor at a recycled address still miss."""
    from pdm_ssd_amd import _native
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    xyz_np = clouds("lidar", 2, 4096, seed=201)
    xyz = T(xyz_np, dev)
    new_xyz = xyz[:, :512].contiguous()
    new_np = xyz_np[:, :512].copy()
    builds = []
    orig = _native.call

    def counting(name, *a):
        if name == "pdm_grid_build":
            builds.append(name)
        return orig(name, *a)
    _native.call = counting
    try:
        assert ext.GRID_CACHE.depth == 0 and not ext.GRID_CACHE.entries
        with pu.shared_search_grids():
            for r, ns in ((0.3, 16), (1.2, 32)):
                got = pu.ball_query(r, ns, xyz, new_xyz)
                np.testing.assert_array_equal(got.cpu().numpy(), oracle.ball_query(r, ns, xyz_np, new_np))
            d, i = pu.three_nn(new_xyz, xyz)                  # known set = the same tensor: third user of the grid
            rd, ri = oracle.three_nn(new_np, xyz_np)
            np.testing.assert_array_equal(i.cpu().numpy(), ri)
            assert len(builds) == 1
            xyz.add_(0.25)                                    # in-place write: version changes, the grid is rebuilt
            moved = xyz_np + np.float32(0.25)
            got = pu.ball_query(0.3, 16, xyz, new_xyz)
            np.testing.assert_array_equal(got.cpu().numpy(), oracle.ball_query(0.3, 16, moved, new_np))
            assert len(builds) == 2
            ptr = xyz.data_ptr()
            del xyz, got
            other_np = clouds("uniform", 2, 4096, seed=202)
            other = T(other_np, dev)                          # usually lands on the address just freed
            got = pu.ball_query(0.8, 16, other, new_xyz)
            np.testing.assert_array_equal(got.cpu().numpy(), oracle.ball_query(0.8, 16, other_np, new_np))
            assert len(builds) == 3, (ptr, other.data_ptr())
        assert not ext.GRID_CACHE.entries                     # nothing survives the scope
        # a static buffer rewritten through a raw pointer between two calls: _version does not move
        static = T(xyz_np, dev)
        got = pu.ball_query(0.3, 16, static, new_xyz)
        np.testing.assert_array_equal(got.cpu().numpy(), oracle.ball_query(0.3, 16, xyz_np, new_np))
        v = static._version
        _native.copy_many([static], [T(other_np, dev)])
        assert static._version == v
        n0 = len(builds)
        got = pu.ball_query(0.8, 16, static, new_xyz)
        np.testing.assert_array_equal(got.cpu().numpy(), oracle.ball_query(0.8, 16, other_np, new_np))
        assert len(builds) == n0 + 1 and not ext.GRID_CACHE.entries
    finally:
        _native.call = orig


@pytest.mark.parametrize("quad,heavy,cpw", [(3, 96, 16), (2, 96, 16), (2, 0, 16), (2, 100000, 32), (2, 7, 64), (2, 96, 64), (1, 96, 16), (0, 96, 16)])
def test_ball_query_grid_quad_and_wave_kernels_agree_with_oracle(oracle, dev, quad, heavy, cpw):
    """One centre per lane (hits kept in index order in a per-lane list; centres with many candidates or rows handed to
    the whole-wave path — threshold 0: every centre takes that path, 100000: none does unless its box has too many rows),
    four centres per wave (DPP rows) and one centre per wave: same indices as the oracle on clouds whose balls hold
    0 .. far more than 16 candidates (dense lidar near field, duplicated points, a lattice, radius 0 and huge)."""
    from pdm_ssd_amd import _native
    old = _native.lib().pdm_tune_bq_quad(quad)
    old_heavy = _native.lib().pdm_tune_bq_heavy(heavy)
    old_cpw = _native.lib().pdm_tune_bq_cpw(cpw)
    try:
        lid = clouds("lidar", 2, 8192, seed=203)
        dup = np.concatenate([lid[:, :4096], lid[:, :4096]], axis=1)
        lattice = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32) * 0.25
        for xyz_np, m in ((lid, 2048), (dup, 1000), (lattice, 4096), (clouds("uniform", 3, 5000, seed=204), 1251)):
            new_np = np.ascontiguousarray(xyz_np[:, :m])
            xyz, new_xyz = T(xyz_np, dev), T(new_np, dev)
            for r, ns in ((0.1, 16), (0.5, 32), (2.0, 16), (0.0, 8), (1e4, 32)):
                got = pu.ball_query(r, ns, xyz, new_xyz)
                np.testing.assert_array_equal(got.cpu().numpy(), oracle.ball_query(r, ns, xyz_np, new_np), err_msg=f"r={r} ns={ns} m={m}")
    finally:
        _native.lib().pdm_tune_bq_quad(old)
        _native.lib().pdm_tune_bq_heavy(old_heavy)
        _native.lib().pdm_tune_bq_cpw(old_cpw)
