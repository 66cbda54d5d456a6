"""PDM neck scatter (build-defined spec): HIP kernels vs oracle/pdm_oracle.c.

fp32 with free summation order (atomics) and device transcendental functions -> tolerance 1e-4
relative to the tensor's scale, as north_star states for features.  Parity is with THIS repo's
written spec only: the reference snapshot has no PDM source.
"""
import numpy as np
import pytest
import torch

from pdm_ssd_amd import pdm_ops

pytestmark = pytest.mark.gpu

RANGE = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)


def make_inputs(B, P, C, degree, seed, outliers=True):
    rng = np.random.default_rng(seed)
    xyz = np.stack([rng.uniform(0, 70.4, (B, P)), rng.uniform(-40, 40, (B, P)), rng.uniform(-3, 1, (B, P))], -1)
    xyz = xyz.astype(np.float32)
    if outliers:
        xyz[:, 0] = [-0.3, -39.9, 0.9]      # dilation block partly outside
        xyz[:, 1] = [500.0, 0.0, 0.0]       # far outside: contributes nothing
        xyz[:, 2] = [np.nan, 0.0, 0.0]      # NaN point: skipped
        xyz[:, 3] = [70.39, 39.99, 0.99]    # last cell
        xyz[:, 4] = xyz[:, 5]               # two centres in the same cell (multi-centre fusion)
    feat = rng.standard_normal((B, P, C)).astype(np.float32)
    sh = (rng.standard_normal((B, P, (degree + 1) ** 2)) * 0.5).astype(np.float32)
    sh[..., 0] += 3.0
    sigma = rng.uniform(0.3, 1.5, (B, P)).astype(np.float32)
    inv2s2 = (0.5 / (sigma * sigma)).astype(np.float32)
    return xyz, feat, sh, inv2s2


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("layout", [1, 0])
@pytest.mark.parametrize("degree", [0, 1, 2, 3])
@pytest.mark.parametrize("cell,kernel", [((0.8, 0.8, 4.0), (7, 7, 1)), ((1.6, 1.6, 2.0), (3, 5, 3))])
def test_scatter_matches_oracle(oracle, dev, layout, degree, cell, kernel):
    B, P, C = 2, 200, 70
    xyz, feat, sh, inv2s2 = make_inputs(B, P, C, degree, seed=degree * 10 + layout)
    g = pdm_ops.BevGrid(RANGE, cell)
    origin, cellf, inv_cell, dims = oracle.pdm_grid_params(RANGE, cell)
    assert dims == (g.W, g.H, g.D)
    np.testing.assert_array_equal(inv_cell, g.inv_cell)
    ref_grid, ref_wsum = oracle.pdm_scatter(xyz, feat, sh, inv2s2, origin, cellf, inv_cell, dims, kernel, degree, layout)
    grid, wsum = pdm_ops.pdm_scatter(T(xyz, dev), T(feat, dev), T(sh, dev), T(inv2s2, dev), g, kernel, degree, layout)
    scale = np.abs(ref_grid).max()
    np.testing.assert_allclose(grid.cpu().numpy(), ref_grid, rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(wsum.cpu().numpy(), ref_wsum, rtol=1e-4, atol=1e-4 * np.abs(ref_wsum).max())
    # sparsity: cells no dilation block touches are exactly zero
    assert (grid.cpu().numpy()[ref_grid == 0] == 0).all()
    # normalisation
    ref_n = oracle.pdm_normalize(ref_grid, ref_wsum, C, dims, layout)
    got_n = pdm_ops.bev_normalize_(grid.clone(), wsum, C, g, layout).cpu().numpy()
    np.testing.assert_allclose(got_n, ref_n, rtol=2e-3, atol=2e-3 * np.abs(ref_n).max())


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
@pytest.mark.parametrize("cell,kernel,C", [((0.8, 0.8, 4.0), (7, 7, 1), 128), ((0.8, 0.8, 4.0), (7, 7, 1), 70),
                                           ((1.6, 1.6, 2.0), (3, 5, 3), 70), ((0.4, 0.4, 4.0), (1, 1, 1), 33),
                                           ((3.2, 1.6, 4.0), (17, 9, 1), 64)])
def test_gather_matches_oracle(oracle, dev, degree, cell, kernel, C):
    """Gather form (inference): same sums as the scatter spec, every cell written, wsum included."""
    B, P = 2, 300
    assert pdm_ops.gather_supported(C, int(round(4.0 / cell[2])))
    xyz, feat, sh, inv2s2 = make_inputs(B, P, C, degree, seed=degree * 7 + C)
    g = pdm_ops.BevGrid(RANGE, cell)
    origin, cellf, inv_cell, dims = oracle.pdm_grid_params(RANGE, cell)
    ref_grid, ref_wsum = oracle.pdm_scatter(xyz, feat, sh, inv2s2, origin, cellf, inv_cell, dims, kernel, degree, 1)
    args = (T(xyz, dev), T(feat, dev), T(sh, dev), T(inv2s2, dev), g, kernel, degree)
    grid, wsum = pdm_ops.pdm_gather(*args, normalize=False)
    scale = np.abs(ref_grid).max()
    np.testing.assert_allclose(grid.cpu().numpy(), ref_grid, rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(wsum.cpu().numpy(), ref_wsum, rtol=1e-4, atol=1e-4 * np.abs(ref_wsum).max())
    assert (grid.cpu().numpy()[ref_grid == 0] == 0).all()
    ref_n = oracle.pdm_normalize(ref_grid, ref_wsum, C, dims, 1)
    got_n, wsum_n = pdm_ops.pdm_gather(*args, normalize=True)
    np.testing.assert_allclose(got_n.cpu().numpy(), ref_n, rtol=2e-3, atol=2e-3 * np.abs(ref_n).max())
    assert torch.equal(wsum_n, wsum)
    # fixed summation order: bitwise reproducible, also into a dirty output buffer of another launch
    again, _ = pdm_ops.pdm_gather(*args, normalize=True)
    assert torch.equal(again, got_n)


@pytest.mark.parametrize("P", [700, 1500])
def test_gather_dense_tile_and_edge_sizes(oracle, dev, P):
    """Many points in one tile (several 32-point chunks; 700: a long list sorted in LDS, 1500: longer than the LDS list of
    csrc/pdm_gather.hip, sorted through the workspace), P = 1, P = 0 and B = 0."""
    rng = np.random.default_rng(5)
    B, C, degree, kernel, cell = 2, 48, 2, (5, 5, 1), (0.8, 0.8, 4.0)
    xyz, feat, sh, inv2s2 = make_inputs(B, P, C, degree, seed=11, outliers=False)
    xyz[0, :, 0] = rng.uniform(10.0, 13.0, P); xyz[0, :, 1] = rng.uniform(-2.0, 1.0, P)   # sample 0: 4 x 4 cells
    g = pdm_ops.BevGrid(RANGE, cell)
    origin, cellf, inv_cell, dims = oracle.pdm_grid_params(RANGE, cell)
    ref_grid, ref_wsum = oracle.pdm_scatter(xyz, feat, sh, inv2s2, origin, cellf, inv_cell, dims, kernel, degree, 1)
    grid, wsum = pdm_ops.pdm_gather(T(xyz, dev), T(feat, dev), T(sh, dev), T(inv2s2, dev), g, kernel, degree, normalize=False)
    np.testing.assert_allclose(grid.cpu().numpy(), ref_grid, rtol=1e-4, atol=1e-4 * np.abs(ref_grid).max())
    np.testing.assert_allclose(wsum.cpu().numpy(), ref_wsum, rtol=1e-4, atol=1e-4 * np.abs(ref_wsum).max())
    for p in (1, 0):
        a = [T(v[:, :p], dev) for v in (xyz, feat, sh, inv2s2)]
        rg, rw = oracle.pdm_scatter(xyz[:, :p], feat[:, :p], sh[:, :p], inv2s2[:, :p], origin, cellf, inv_cell, dims, kernel, degree, 1)
        gg, ww = pdm_ops.pdm_gather(*a, g, kernel, degree, normalize=False)
        np.testing.assert_allclose(gg.cpu().numpy(), rg, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ww.cpu().numpy(), rw, rtol=1e-4, atol=1e-6)
    a = [T(v[:0], dev) for v in (xyz, feat, sh, inv2s2)]
    gg, ww = pdm_ops.pdm_gather(*a, g, kernel, degree)
    assert gg.shape[0] == 0 and ww.shape[0] == 0


def test_gather_rejects_what_it_cannot_hold(dev):
    g = pdm_ops.BevGrid(RANGE, (0.8, 0.8, 0.5))   # D = 8, C = 128: accumulator tile exceeds LDS
    assert not pdm_ops.gather_supported(128, 8)
    xyz, feat, sh, inv2s2 = make_inputs(1, 16, 128, 1, seed=0)
    with pytest.raises(RuntimeError, match="LDS"):
        pdm_ops.pdm_gather(T(xyz, dev), T(feat, dev), T(sh, dev), T(inv2s2, dev), g, (3, 3, 3), 1)


def test_neck_eval_gather_equals_scatter_path(dev):
    """PDMNeck inference (gather form) against its own scatter + normalise path on the same weights."""
    from pdm_ssd_amd.pdm_neck import PDMNeck
    cfg = {'SOURCE_LAYER': 1, 'FEATURE_DIM': 128, 'DILATION': [7, 7, 1], 'SH_DEGREE': 2, 'BEV_STRIDE': 8,
           'HEIGHT_BINS': 1, 'INPUT_CHANNELS': 40}
    torch.manual_seed(0)
    neck = PDMNeck(cfg, grid_size=[1408, 1600, 40], voxel_size=[0.05, 0.05, 0.1], point_cloud_range=RANGE).to(dev).eval()
    with torch.no_grad():
        neck.coef.weight.normal_(0.0, 0.05)
    xyz, feat, _, _ = make_inputs(3, 1024, 40, 2, seed=9)
    bd = {'sa_xyz': [None, T(xyz, dev)], 'sa_features': [None, T(feat, dev).transpose(1, 2).contiguous()]}
    with torch.no_grad():
        a = neck(dict(bd))
        assert neck._pdm_fused_cache['proj'][1] is not None and neck._pdm_fused_cache['coef'][1] is not None
        neck.use_gather = False
        neck.use_fused = False     # torch Conv1d/BatchNorm1d for the projection and coefficient heads
        b = neck(dict(bd))
    assert a['spatial_features'].shape == b['spatial_features'].shape == (3, 128, 200, 176)
    sa, sb = a['spatial_features'], b['spatial_features']
    assert torch.allclose(sa, sb, rtol=2e-3, atol=2e-3 * float(sb.abs().max()))
    assert torch.allclose(a['pdm_weight_sum'], b['pdm_weight_sum'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("C,degree,kernel", [(128, 2, (7, 7, 1)), (70, 3, (5, 7, 1)), (200, 1, (3, 3, 1))])
def test_training_gather_form_matches_scatter_and_normalize_autograd(dev, C, degree, kernel):
    """pdm_gather_normalized (the neck's training form: gather kernel forward; backward = dL/dwsum pass + the gradient kernel
    dividing by wsum on its own reads, no dL/dgrid tensor) against PDMScatter -> BevNormalize under autograd, whose backward the
    oracle tests pin: the normalised map, the weight sums and the gradients of feat, sh and inv2s2 (which carry dL/dwsum)."""
    B, P = 2, 600
    xyz, feat, sh, inv2s2 = make_inputs(B, P, C, degree, seed=21)
    sh[0, :40, :] *= 1e-4                                      # cells whose weight sum stays under eps: y = grid there
    g = pdm_ops.BevGrid(RANGE, (0.8, 0.8, 4.0))
    gy = torch.randn(B, g.H, g.W, C, generator=torch.Generator().manual_seed(3)).to(dev)
    res = []
    for form in ("gather", "scatter"):
        a = [T(v, dev).requires_grad_(i > 0) for i, v in enumerate((xyz, feat, sh, inv2s2))]
        if form == "gather":
            y, w = pdm_ops.pdm_gather_normalized(*a, g, kernel, degree, 1e-3)
        else:
            grid, w = pdm_ops.pdm_scatter(*a, g, kernel, degree, 1)
            y = pdm_ops.bev_normalize(grid, w, C, g, 1e-3)
        (y.view(B, g.H, g.W, C) * gy).sum().backward()
        res.append((y.detach().view(B, g.H, g.W, C), w.detach().view(B, g.H, g.W), [t.grad for t in a[1:]]))
    (ya, wa, ga), (yb, wb, gb) = res
    assert float((wb.abs() <= 1e-3).float().mean()) > 0.05 and float((wb.abs() > 1e-3).float().mean()) > 0.05
    torch.testing.assert_close(wa, wb, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ya, yb, rtol=1e-4, atol=1e-4 * float(yb.abs().max()))
    for name, x, r in zip(("dfeat", "dsh", "dinv2s2"), ga, gb):
        assert float((x - r).abs().max()) <= 2e-4 * float(r.abs().max()), name


def test_scatter_linearity_and_zero_features(dev):
    """Size-independent property: the scatter is linear in the features."""
    B, P, C = 3, 512, 128
    xyz, feat, sh, inv2s2 = make_inputs(B, P, C, 2, seed=5)
    g = pdm_ops.BevGrid(RANGE, (0.4, 0.4, 4.0))
    args = (T(xyz, dev), None, T(sh, dev), T(inv2s2, dev), g, (7, 7, 1), 2, 1)
    f1 = T(feat, dev)
    f2 = torch.randn_like(f1)
    ga, wa = pdm_ops.pdm_scatter(args[0], f1, *args[2:])
    gb, wb = pdm_ops.pdm_scatter(args[0], f2, *args[2:])
    gc, wc = pdm_ops.pdm_scatter(args[0], 2.0 * f1 - 3.0 * f2, *args[2:])
    torch.testing.assert_close(gc, 2.0 * ga - 3.0 * gb, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(wa, wc, rtol=1e-5, atol=1e-5)
    gz, _ = pdm_ops.pdm_scatter(args[0], torch.zeros_like(f1), *args[2:])
    assert float(gz.abs().max()) == 0.0


@pytest.mark.parametrize("layout", [1, 0])
@pytest.mark.parametrize("degree,kernel,cell", [(2, (7, 7, 1), (0.8, 0.8, 4.0)), (3, (3, 3, 3), (1.6, 1.6, 2.0))])
def test_scatter_grad_matches_oracle(oracle, dev, layout, degree, kernel, cell):
    B, P, C = 2, 150, 70
    xyz, feat, sh, inv2s2 = make_inputs(B, P, C, degree, seed=77)
    g = pdm_ops.BevGrid(RANGE, cell)
    origin, cellf, inv_cell, dims = oracle.pdm_grid_params(RANGE, cell)
    rng = np.random.default_rng(1)
    shape = (B, g.H, g.W, C * g.D) if layout == 1 else (B, C * g.D, g.H, g.W)
    dgrid = rng.standard_normal(shape).astype(np.float32)
    dwsum = rng.standard_normal((B, g.H, g.W, g.D)).astype(np.float32)
    rf, rs, ri = oracle.pdm_scatter_grad(xyz, feat, sh, inv2s2, origin, cellf, inv_cell, dims, kernel, degree,
                                         dgrid, dwsum, layout)
    f, s, i2 = (T(a, dev).requires_grad_(True) for a in (feat, sh, inv2s2))
    grid, wsum = pdm_ops.pdm_scatter(T(xyz, dev), f, s, i2, g, kernel, degree, layout)
    torch.autograd.backward([grid, wsum], [T(dgrid, dev), T(dwsum, dev)])
    for got, ref in ((f.grad, rf), (s.grad, rs), (i2.grad, ri)):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(ref).max()))


def test_neck_module_contract(dev):
    from pdm_ssd_amd.pdm_neck import PDMNeck
    cfg = {'SOURCE_LAYER': 1, 'FEATURE_DIM': 32, 'DILATION': [5, 5, 1], 'SH_DEGREE': 2, 'BEV_STRIDE': 8,
           'HEIGHT_BINS': 2, 'INPUT_CHANNELS': 24}
    neck = PDMNeck(cfg, grid_size=[1408, 1600, 40], voxel_size=[0.05, 0.05, 0.1], point_cloud_range=RANGE).to(dev)
    assert neck.num_bev_features == 64
    xyz, feat, _, _ = make_inputs(2, 300, 24, 2, seed=3, outliers=False)
    bd = {'sa_xyz': [None, T(xyz, dev)], 'sa_features': [None, T(feat, dev).transpose(1, 2).contiguous()]}
    neck.train()
    out = neck(bd)
    sf = out['spatial_features']
    assert tuple(sf.shape) == (2, 64, 200, 176) and out['spatial_features_stride'] == 8
    sf.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in neck.parameters())
    neck.eval()
    with torch.no_grad():
        sf2 = neck({'sa_xyz': bd['sa_xyz'], 'sa_features': bd['sa_features']})['spatial_features']
    assert torch.isfinite(sf2).all()


@pytest.mark.parametrize("D,C", [(1, 128), (2, 32), (3, 5)])
def test_bev_normalize_function_matches_torch_autograd(dev, D, C):
    """BevNormalize (one kernel forward, one backward) against the torch expression it replaces, values and both gradients."""
    torch.manual_seed(D * 10 + C)
    g = pdm_ops.BevGrid(RANGE, (3.2, 3.2, 4.0 / D))
    B = 2
    x0 = torch.randn(B, g.H, g.W, C * D, device=dev)
    w0 = torch.randn(B, g.H, g.W, D, device=dev)
    w0[torch.rand_like(w0) < 0.3] = 0.0                       # untouched cells: passed through
    dy = torch.randn_like(x0)
    x1, w1 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    w5 = w1.unsqueeze(-2)
    x5 = x1.view(B, g.H, g.W, C, D)
    ref = torch.where(w5.abs() > 1e-6, x5 / torch.where(w5.abs() > 1e-6, w5, torch.ones_like(w5)), x5).view_as(x1)
    ref.backward(dy)
    x2, w2 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    out = pdm_ops.bev_normalize(x2 * 1.0, w2, C, g)           # (x2 * 1.0: the Function consumes its input in place)
    out.backward(dy)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(x2.grad, x1.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(w2.grad, w1.grad, rtol=1e-4, atol=1e-4)
