"""Neighbour-list compaction (pdm_sa_pack / pdm_sa_mlp_packed, csrc/sa_pack.hip): the row list is checked against a
numpy restatement of its contract, and the SA kernels over it must be BIT-identical to the same kernels over the
dense (B,M,nsample) list — dropping padding copies of the first hit does not change a max-pool."""
import copy

import numpy as np
import pytest
import torch

from pdm_ssd_amd import fused, synthetic
from pdm_ssd_amd.pointnet2_batch import pointnet2_modules as pm
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu

pytestmark = pytest.mark.gpu


def expected_pack(idx, n):
    """numpy restatement of the layout documented in include/pdmssd_hip.h (pdm_sa_pack)."""
    B, M, ns = idx.shape
    flat = idx.reshape(B * M, ns)
    diff = flat != flat[:, :1]
    cnt = np.where(diff.any(1), ns - np.argmax(diff[:, ::-1], axis=1), 1)
    cls = np.where(cnt <= 1, 0, np.ceil(np.log2(np.maximum(cnt, 1))).astype(np.int64))
    rows, meta, r = [], [], 0
    for k in range(6):
        align = 32 if k == 5 else 16
        pad = (-r) % align
        rows += [(0, -1)] * pad
        r += pad
        meta.append(r)
        for c in np.nonzero(cls == k)[0]:
            b = c // M
            for s in range(1 << k):
                rows.append((b * n + (flat[c, s] if s < cnt[c] else flat[c, 0]), c))
            r += 1 << k
    pad = (-r) % 32
    rows += [(0, -1)] * pad
    live = int(((1 << cls)).sum())
    return np.array(rows, dtype=np.int32).reshape(-1, 2), meta + [r + pad, live]


@pytest.mark.parametrize("ns,radius,n,m", [(16, 0.6, 2000, 300), (32, 1.2, 2000, 300), (32, 6.0, 600, 70), (16, 0.01, 500, 9)])
def test_pack_layout(dev, ns, radius, n, m):
    cl = synthetic.lidar_like_clouds(3, n, 21)
    xyz = torch.from_numpy(np.ascontiguousarray(cl[:, :, :3])).to(dev)
    new_xyz = xyz[:, :m].contiguous()
    idx = pu.ball_query(radius, ns, xyz, new_xyz)
    pack, meta = fused.sa_pack(idx, n)
    want_rows, want_meta = expected_pack(idx.cpu().numpy(), n)
    got_meta = meta.cpu().numpy().tolist()
    assert got_meta == want_meta
    np.testing.assert_array_equal(pack.cpu().numpy()[:want_meta[6]], want_rows)


def test_pack_arbitrary_index_tensor(dev):
    """Not a ball_query result: duplicates of slot 0 in the middle stay, trailing ones go."""
    idx = torch.tensor([[[5, 7, 5, 9] + [5] * 12, [3] * 16, [1, 2] + [1] * 14, list(range(16))]], dtype=torch.int32, device=dev)
    pack, meta = fused.sa_pack(idx, 40)
    want_rows, want_meta = expected_pack(idx.cpu().numpy(), 40)
    assert meta.cpu().numpy().tolist() == want_meta
    np.testing.assert_array_equal(pack.cpu().numpy()[:want_meta[6]], want_rows)


@pytest.mark.parametrize("cin,mlps,nsamples,radii,npoint", [
    (1, [[1, 16, 16, 32], [1, 32, 32, 64]], [16, 32], [0.5, 1.5], 400),       # register-resident form (SA1 shapes)
    (96, [[96, 64, 64, 128], [96, 64, 96, 128]], [16, 32], [0.9, 3.0], 300),  # hoisted, W = 2
    (128, [[128, 128, 196, 256], [128, 128, 196, 256]], [16, 32], [1.0, 4.0], 150),   # W = 4, two-tile groups
    (32, [[32, 256, 384, 512]], [32], [5.0], 64),                             # W = 8
    (6, [[6, 20, 36]], [32], [2.5], 100),                                     # unhoisted chain form, odd widths
    (0, [[0, 64]], [16], [0.3], 37),                                          # xyz only, single layer, ragged tail
])
def test_packed_sa_is_bit_identical_to_dense(dev, cin, mlps, nsamples, radii, npoint):
    torch.manual_seed(cin + 3)
    sa = pm.PointnetSAModuleMSG(npoint=npoint, radii=radii, nsamples=nsamples, mlps=copy.deepcopy(mlps)).eval().to(dev)
    cl = synthetic.lidar_like_clouds(2, 1800, 5)
    xyz = torch.from_numpy(np.ascontiguousarray(cl[:, :, :3])).to(dev)
    feat = torch.randn(2, cin, 1800, device=dev) if cin else None
    with torch.no_grad():
        new_xyz = sa.sample(xyz)
        packs = sa.query(xyz, new_xyz)
        assert all(isinstance(p, tuple) for p in packs)
        # the clouds must exercise several segment classes, otherwise the comparison proves little
        classes = set()
        for p in packs:
            mt = p[1].cpu().numpy()
            classes |= {k for k in range(6) if (mt[k + 1] if k < 5 else mt[6]) > mt[k]}
        _, got = sa(xyz, feat, new_xyz=new_xyz, idx_list=packs)
        got = got.clone()
        sa.use_pack = False
        dense = sa.query(xyz, new_xyz)
        assert all(torch.is_tensor(d) for d in dense)
        _, want = sa(xyz, feat, new_xyz=new_xyz, idx_list=dense)
    assert len(classes) >= 2, classes
    assert torch.equal(got, want)


@pytest.mark.parametrize("cin,mlps,npoint,hoist", [
    (96, [[96, 64, 64, 128], [96, 64, 96, 128]], 300, True),          # SA2 of PointNet2MSG
    (256, [[256, 128, 196, 256], [256, 128, 196, 256]], 150, True),   # SA3
    (512, [[512, 256, 256, 512], [512, 256, 384, 512]], 64, True),    # SA4: 64 centres per cloud, a few tile pairs per scale
    (512, [[512, 256, 256, 512], [512, 256, 384, 512]], 64, False),   # ... unhoisted
    (1, [[1, 16, 16, 32], [1, 32, 32, 64]], 400, False),              # SA1: register kernels, two launches inside the entry point
])
def test_both_scales_in_one_launch_equal_two_launches(dev, cin, mlps, npoint, hoist):
    """pdm_sa_mlp_packed_pair (both scales of an MSG level, blockIdx.y = scale) against one pdm_sa_mlp_packed launch per scale,
    and against the entry point's own two-launch route: bit-identical pooled features."""
    from pdm_ssd_amd import _native
    torch.manual_seed(cin + npoint)
    sa = pm.PointnetSAModuleMSG(npoint=npoint, radii=[1.0, 3.0], nsamples=[16, 32], mlps=copy.deepcopy(mlps)).eval().to(dev)
    sa.use_pre = hoist
    cl = synthetic.lidar_like_clouds(3, 1800, 9)
    xyz = torch.from_numpy(np.ascontiguousarray(cl[:, :, :3])).to(dev)
    feat = torch.randn(3, cin, 1800, device=dev)
    calls = []
    orig = _native.call
    _native.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        with torch.no_grad():
            new_xyz = sa.sample(xyz)
            packs = sa.query(xyz, new_xyz)
            _, one = sa(xyz, feat, new_xyz=new_xyz, idx_list=packs)
            one = one.clone()
            assert calls.count("pdm_sa_mlp_packed_pair") == 1 and calls.count("pdm_sa_mlp_packed") == 0
            sa.use_pair = False
            _, two = sa(xyz, feat, new_xyz=new_xyz, idx_list=packs)
            two = two.clone()
            assert calls.count("pdm_sa_mlp_packed") == 2
            sa.use_pair = True
            old = _native.lib().pdm_tune_sa_pair(0)
            _, three = sa(xyz, feat, new_xyz=new_xyz, idx_list=packs)
            _native.lib().pdm_tune_sa_pair(old)
    finally:
        _native.call = orig
    assert torch.isfinite(one).all() and torch.equal(one, two) and torch.equal(one, three)


def test_pack_pair_equals_two_single_packs(dev):
    """pdm_sa_pack_pair (both scales of a level in one count -> scan -> fill sequence) == pdm_sa_pack per scale: meta and every
    live row identical, for different nsample per scale and a centre count that is not a multiple of the workgroup size."""
    cl = synthetic.lidar_like_clouds(3, 2500, 4)
    xyz = torch.from_numpy(np.ascontiguousarray(cl[:, :, :3])).to(dev)
    new_xyz = xyz[:, :333].contiguous()
    i0 = pu.ball_query(0.6, 16, xyz, new_xyz)
    i1 = pu.ball_query(2.5, 32, xyz, new_xyz)
    (p0, m0), (p1, m1) = fused.sa_pack_pair(i0, i1, xyz.shape[1])
    for idx, p, mt in ((i0, p0, m0), (i1, p1, m1)):
        ps, ms = fused.sa_pack(idx, xyz.shape[1])
        assert torch.equal(ms, mt)
        rows = int(mt[6])
        assert rows > 0 and torch.equal(ps[:rows], p[:rows])


def test_packed_sa_empty_and_tiny(dev):
    sa = pm.PointnetSAModuleMSG(npoint=3, radii=[0.5], nsamples=[32], mlps=[[4, 16, 32]]).eval().to(dev)
    xyz = torch.rand(1, 50, 3, device=dev)
    feat = torch.randn(1, 4, 50, device=dev)
    with torch.no_grad():
        _, got = sa(xyz, feat)
        sa.use_pack = False
        _, want = sa(xyz, feat)
    assert got.shape == (1, 32, 3) and torch.equal(got, want)
    pack, meta = fused.sa_pack(torch.zeros((0, 5, 16), dtype=torch.int32, device=dev), 10)
    assert meta.cpu().numpy().tolist() == [0] * 8
