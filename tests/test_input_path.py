"""Input path: the build-defined draw of sample_points.
CPU: the numpy oracle satisfies the reference's rule (data_processor.py:189-210) property by property.
GPU: the HIP kernel reproduces the oracle row for row (the draw is a deterministic hash, so parity is exact)."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import cpu_oracle as o


def lidar_raw(n, seed, far_frac=0.2):
    rng = np.random.default_rng(seed)
    r = np.where(rng.uniform(size=n) < far_frac, rng.uniform(40.0, 70.0, n), rng.uniform(1.0, 39.9, n))
    th = rng.uniform(-np.pi, np.pi, n)
    z = rng.uniform(-3, 1, n)
    rho = np.sqrt(np.maximum(r * r - z * z, 0.0))
    return np.stack([rho * np.cos(th), rho * np.sin(th), z, rng.uniform(0, 1, n)], 1).astype(np.float32)


@pytest.mark.parametrize("n,P,far_frac", [(5000, 1024, 0.1), (5000, 1024, 0.5), (900, 1024, 0.2), (1024, 1024, 0.3),
                                          (20000, 16384, 0.05), (513, 1024, 0.0)])
def test_spec_obeys_the_reference_rule(n, P, far_frac):
    pts = lidar_raw(n, n + P, far_frac)
    ch = o.sample_points_choice(pts, P, seed=3, cloud=1)
    depth = np.linalg.norm(pts[:, :3], axis=1)
    far = np.where(~(depth < 40.0))[0]
    assert ch.shape == (P,) and ch.min() >= 0 and ch.max() < n
    if P < n:
        assert len(set(ch.tolist())) == P                                   # without replacement
        if P > len(far):
            assert set(far.tolist()) <= set(ch.tolist())                    # every far point kept (:197-200)
            assert (depth[np.setdiff1d(ch, far)] < 40.0).all()
    else:
        cnt = np.bincount(ch, minlength=n)
        assert cnt.min() >= 1 and cnt.max() <= 2 and cnt.sum() == P         # all points + distinct extras (:205-209)
    assert not np.array_equal(ch, np.sort(ch)) or P < 3                     # shuffled
    # deterministic in (seed, cloud); different for another seed or cloud index
    np.testing.assert_array_equal(ch, o.sample_points_choice(pts, P, seed=3, cloud=1))
    assert not np.array_equal(ch, o.sample_points_choice(pts, P, seed=4, cloud=1))
    assert not np.array_equal(ch, o.sample_points_choice(pts, P, seed=3, cloud=2))


def test_draw_is_close_to_uniform():
    """Near points are drawn (almost) uniformly: each of 4000 near points is picked about 25 % of the time."""
    pts = lidar_raw(4000, 1, far_frac=0.0)
    hits = np.zeros(4000)
    for seed in range(200):
        hits[o.sample_points_choice(pts, 1000, seed, 0)] += 1
    assert abs(hits.mean() - 50.0) < 1e-9 and hits.std() < 8.0 and hits.min() > 20 and hits.max() < 85


def test_bin_reader_roundtrip():
    from pdm_ssd_amd.input_path import read_velodyne_bin
    pts = lidar_raw(1234, 9)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "000001.bin")
        pts.tofile(path)
        np.testing.assert_array_equal(read_velodyne_bin(path), pts)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [16384, 1024, 1000])
def test_kernel_matches_oracle_rows(dev, P):
    from pdm_ssd_amd import input_path as ip
    sizes = [120000, P // 2 + 1, P, P + 1, 30000, 3 * P]
    clouds = [lidar_raw(n, 100 + i, far_frac=[0.1, 0.2, 0.3, 0.0, 0.9, 0.4][i]) for i, n in enumerate(sizes)]
    raw, cnt, host = ip.upload_raw(clouds, dev)
    out, choice = ip.sample_points_batch(raw, cnt, P, seed=11, host_counts=host, return_choice=True)
    ref_rows, ref_choice = o.sample_points_batch(clouds, P, 11)
    np.testing.assert_array_equal(choice.cpu().numpy(), ref_choice)
    np.testing.assert_array_equal(out.cpu().numpy(), ref_rows)
    assert out.shape == (len(sizes) * P, 5)


@pytest.mark.gpu
def test_sampled_batch_feeds_the_backbone(dev):
    """raw clouds -> GPU sampler -> PointNet2MSG.forward: the collate contract (column 0 = sample index, equal counts)."""
    from pdm_ssd_amd import input_path as ip
    from pdm_ssd_amd.pointnet2_backbone import PointNet2MSG
    cfg = {'SA_CONFIG': {'NPOINTS': [128, 32], 'RADIUS': [[0.5, 1.0], [1.0, 2.0]], 'NSAMPLE': [[16, 32], [16, 32]],
                         'MLPS': [[[16, 16], [16, 32]], [[32, 32], [32, 64]]]}, 'FP_MLPS': [[32, 32], [64, 64]]}
    net = PointNet2MSG(cfg, input_channels=4).to(dev).eval()
    clouds = [lidar_raw(n, n) for n in (5000, 900, 2048)]
    raw, cnt, host = ip.upload_raw(clouds, dev)
    pts = ip.sample_points_batch(raw, cnt, 1024, seed=1, host_counts=host)
    with torch.no_grad():
        bd = net({'batch_size': 3, 'points': pts})
    assert bd['point_features'].shape == (3 * 1024, 32) and torch.isfinite(bd['point_features']).all()


@pytest.mark.gpu
def test_sampler_rejects_what_the_reference_rejects(dev):
    from pdm_ssd_amd import input_path as ip
    raw, cnt, host = ip.upload_raw([lidar_raw(100, 0)], dev)
    with pytest.raises(ValueError):
        ip.sample_points_batch(raw, cnt, 1024, host_counts=host)            # 924 extra picks from 100 points
    with pytest.raises(ValueError):
        ip.sample_points_batch(raw, cnt, 20000)
