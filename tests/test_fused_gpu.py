"""Fused fp32-MFMA inference kernels (pdm_sa_mlp_fused / pdm_fp_mlp_fused) against the unfused graph:
CPU oracle operators + torch-CPU Conv/BN/ReLU/max-pool.  fp32 with a different summation order ->
1e-4, the tolerance north_star states for features."""
import copy

import numpy as np
import pytest
import torch

from pdm_ssd_amd import fused, synthetic
from pdm_ssd_amd.pointnet2_batch import pointnet2_modules as pm

pytestmark = pytest.mark.gpu


def randomize_bn(module, seed):
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


@pytest.mark.parametrize("cin,mlps,nsamples", [
    (1, [[1, 16, 16, 32], [1, 32, 32, 64]], [16, 32]),            # SA1 shapes (K0 = 4)
    (96, [[96, 64, 64, 128], [96, 64, 96, 128]], [16, 32]),       # SA2
    (24, [[24, 128, 196, 256]], [48]),                            # 196 -> padded 208, 3 tiles per centre
    (0, [[0, 20, 36]], [16]),                                     # no features, 2 layers, odd widths
    (8, [[8, 64]], [32]),                                         # single layer
    (12, [[12, 32, 48, 64, 32]], [16]),                           # four layers
])
def test_sa_fused_matches_cpu_graph(dev, cin, mlps, nsamples):
    from oracle import cpu_backbone
    torch.manual_seed(cin + 1)
    radii = [0.9, 1.8][:len(mlps)]
    sa = pm.PointnetSAModuleMSG(npoint=200, radii=radii, nsamples=nsamples, mlps=copy.deepcopy(mlps)).eval()
    randomize_bn(sa, 5)
    cl = synthetic.lidar_like_clouds(2, 1500, 11)
    xyz = np.ascontiguousarray(cl[:, :, :3])
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((2, cin, 1500)).astype(np.float32) if cin else None
    ref_xyz, ref_feat = cpu_backbone.sa_forward(sa, xyz, feat)
    sa_g = copy.deepcopy(sa).to(dev)
    with torch.no_grad():
        nx, nf = sa_g(torch.from_numpy(xyz).to(dev), None if feat is None else torch.from_numpy(feat).to(dev))
    assert '_pdm_fused_cache' in sa_g.__dict__ and all(v[1] is not None for v in sa_g._pdm_fused_cache.values())
    assert nf.stride(1) == 1, "fused output must be a view of point-major storage"
    np.testing.assert_array_equal(nx.cpu().numpy(), ref_xyz)
    np.testing.assert_allclose(nf.cpu().numpy(), ref_feat, rtol=1e-4, atol=1e-4)
    assert ('pre' in sa_g._pdm_fused_cache) == (cin >= pm.PRE_MIN_CIN)
    # first-layer hoisting switched off (features gathered and contracted per pair), then the whole fused path off
    sa_g.use_pre = False
    with torch.no_grad():
        _, nf1 = sa_g(torch.from_numpy(xyz).to(dev), None if feat is None else torch.from_numpy(feat).to(dev))
    np.testing.assert_allclose(nf1.cpu().numpy(), ref_feat, rtol=1e-4, atol=1e-4)
    sa_g.use_fused = False
    with torch.no_grad():
        _, nf2 = sa_g(torch.from_numpy(xyz).to(dev), None if feat is None else torch.from_numpy(feat).to(dev))
    np.testing.assert_allclose(nf.cpu().numpy(), nf2.cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("ck,cs,mlp,n,m", [
    (1024, 512, [512, 512], 256, 64),     # FP4 shapes: input streamed, not staged
    (256, 1, [128, 128], 1000, 300),      # FP1 shapes: c_skip = 1, n not a multiple of 16
    (40, 0, [64, 32], 333, 50),           # no skip features
    (36, 6, [48], 100, 20),               # single layer, unaligned skip rows
    (256, 1, [128, 128], 16390, 4096),    # FP1 at full size: the register-resident chain (rows_chain.hip), tiles across clouds
    (512, 96, [256, 256], 16390, 1024),   # FP2 widths over enough rows for the chain: skip GEMM with a partial k-group
])
def test_fp_fused_matches_cpu_graph(dev, ck, cs, mlp, n, m):
    from oracle import cpu_backbone
    torch.manual_seed(ck)
    fp = pm.PointnetFPModule(mlp=[ck + cs] + mlp).eval()
    randomize_bn(fp, 9)
    rng = np.random.default_rng(1)
    unknown = synthetic.uniform_clouds(2, n, 5)[:, :, :3].copy()
    known = np.ascontiguousarray(unknown[:, :m])
    uf = rng.standard_normal((2, cs, n)).astype(np.float32) if cs else None
    kf = rng.standard_normal((2, ck, m)).astype(np.float32)
    ref = cpu_backbone.fp_forward(fp, unknown, known, uf, kf)
    fp_g = copy.deepcopy(fp).to(dev)
    with torch.no_grad():
        got = fp_g(torch.from_numpy(unknown).to(dev), torch.from_numpy(known).to(dev),
                   None if uf is None else torch.from_numpy(uf).to(dev), torch.from_numpy(kf).to(dev))
    assert got.stride(1) == 1 and fp_g._pdm_fused_cache[0][1] is not None
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    assert fp_g._pdm_fused_cache['pre'][1] is not None      # known-feature block applied on the m known points
    fp_g.use_pre = False                                     # interpolate-then-contract form of the same module
    with torch.no_grad():
        got1 = fp_g(torch.from_numpy(unknown).to(dev), torch.from_numpy(known).to(dev),
                    None if uf is None else torch.from_numpy(uf).to(dev), torch.from_numpy(kf).to(dev))
    np.testing.assert_allclose(got1.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,cin,widths,relu_last", [(1000, 96, [128], False), (37, 515, [256, 64], True),
                                                        (4096, 1, [16], False), (0, 8, [16], True)])
def test_rows_mlp_matches_torch(dev, rows, cin, widths, relu_last):
    """pdm_rows_mlp_fused against torch fp64 Conv1d/BN/ReLU on CPU (1e-4, fp32 accumulation order differs)."""
    torch.manual_seed(rows + cin)
    chans = [cin] + widths
    seq = torch.nn.Sequential(*[m for i in range(len(widths)) for m in
                                (torch.nn.Conv1d(chans[i], chans[i + 1], 1, bias=False),
                                 torch.nn.BatchNorm1d(chans[i + 1]), torch.nn.ReLU())]).eval()
    randomize_bn(seq, 2)
    x = torch.randn(max(rows, 1), cin)
    with torch.no_grad():
        y = x.double().t().unsqueeze(0)
        mods = list(seq.double())
        for i, m in enumerate(mods):
            if i == len(mods) - 1 and not relu_last:
                break
            y = m(y)
        ref = y[0].t().float()[:rows]
    x = x[:rows].contiguous()
    seq.float()
    pk = fused.PackedMLP(fused.split_shared_mlp(seq), dev)
    out = torch.full((rows, pk.dims[-1] + 4), 7.0, device=dev)
    fused.rows_forward(pk, x.to(dev), out, relu_last=relu_last)
    torch.testing.assert_close(out[:, :widths[-1]].cpu(), ref, rtol=1e-4, atol=1e-4)
    assert (out[:, widths[-1]:] == 7.0).all()           # only cout columns are written


def test_pack_cache_follows_weight_updates(dev):
    torch.manual_seed(3)
    fp = pm.PointnetFPModule(mlp=[32, 32]).to(dev).eval()
    unknown = torch.rand(1, 64, 3, device=dev)
    known = unknown[:, :16].contiguous()
    kf = torch.randn(1, 32, 16, device=dev)
    with torch.no_grad():
        a = fp(unknown, known, None, kf).clone()
        fp.mlp[0].weight.mul_(2.0)  # in-place update bumps the version counter -> repack
        b = fp(unknown, known, None, kf)
        fp.use_fused = False
        c = fp(unknown, known, None, kf)
    assert not torch.allclose(a, b)
    torch.testing.assert_close(b.contiguous(), c.contiguous(), rtol=1e-4, atol=1e-4)


def test_training_mode_uses_autograd_path(dev):
    sa = pm.PointnetSAModuleMSG(npoint=64, radii=[1.0], nsamples=[16], mlps=[[4, 16, 16]]).to(dev).train()
    xyz = torch.rand(2, 512, 3, device=dev) * 10
    f = torch.randn(2, 4, 512, device=dev, requires_grad=True)
    _, out = sa(xyz, f)
    out.sum().backward()
    assert f.grad is not None and '_pdm_fused_cache' not in sa.__dict__
