"""pdm_topk_sampling (score-ranked sampling) against the numpy oracle: bit-exact indices, including ties,
signed zeros, infinities and NaNs; the instance-aware SA layer and class_agnostic_nms built on it."""
import copy

import numpy as np
import pytest
import torch

from pdm_ssd_amd import synthetic
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu

pytestmark = pytest.mark.gpu


def scores_case(kind, B, N, seed):
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.standard_normal((B, N)).astype(np.float32)
    if kind == "ties":        # few distinct values: the index order decides almost everything
        return (rng.integers(0, 7, size=(B, N)) / 4.0 - 0.5).astype(np.float32)
    if kind == "sigmoid":     # saturating scores as the SA layer produces them
        return (1.0 / (1.0 + np.exp(-rng.standard_normal((B, N)) * 12.0))).astype(np.float32)
    s = rng.standard_normal((B, N)).astype(np.float32)
    sp = rng.integers(0, N, size=(B, 24))
    vals = np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45, -1e-45], dtype=np.float32)
    for b in range(B):
        s[b, sp[b]] = np.tile(vals, 3)
    return s


@pytest.mark.parametrize("kind", ["random", "ties", "sigmoid", "special"])
@pytest.mark.parametrize("B,N,k", [(3, 16384, 4096), (2, 4096, 1024), (2, 1000, 1000), (1, 777, 1), (2, 65536, 16384), (4, 300, 64)])
def test_topk_matches_oracle(dev, kind, B, N, k):
    from oracle import cpu_oracle as oracle
    s = scores_case(kind, B, N, N + k)
    got = pu.topk_sample(torch.from_numpy(s).to(dev), k).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.topk_sampling(s, k))


def test_topk_errors_and_empty(dev):
    from pdm_ssd_amd import _native
    s = torch.zeros(2, 10, device=dev)
    with pytest.raises(_native.NativeLibraryError):
        pu.topk_sample(s, 11)
    assert pu.topk_sample(s, 0).shape == (2, 0)
    assert pu.topk_sample(s, 10).cpu().numpy().tolist() == [list(range(10))] * 2   # all equal: index order


def test_instance_aware_sa_layer(dev):
    """'cls_aware' sampling = gather of the oracle's top-k indices, then the ordinary SA forward on those centres."""
    from oracle import cpu_oracle as oracle
    from pdm_ssd_amd.instance_aware import PointnetSAModuleMSG_WithSampling
    torch.manual_seed(3)
    sa = PointnetSAModuleMSG_WithSampling(npoint=128, sample_type='cls_aware', radii=[0.8, 1.6], nsamples=[16, 32],
                                          mlps=[[4, 16, 32], [4, 16, 32]], confidence_mlp=[32], num_class=3).to(dev).eval()
    cl = synthetic.lidar_like_clouds(2, 2048, 8)
    xyz = torch.from_numpy(np.ascontiguousarray(cl[:, :, :3])).to(dev)
    feat = torch.randn(2, 4, 2048, device=dev)
    logits = torch.randn(2, 2048, 3, device=dev) * 4
    with torch.no_grad():
        new_xyz, new_feat, cls_preds = sa(xyz, feat, cls_features=logits)
        score = torch.sigmoid(logits.max(dim=-1)[0]).cpu().numpy()
        want_idx = oracle.topk_sampling(score, 128)
        want_xyz = np.take_along_axis(xyz.cpu().numpy(), want_idx[:, :, None].astype(np.int64), 1)
        np.testing.assert_array_equal(new_xyz.cpu().numpy(), want_xyz)
        base_xyz, base_feat = super(PointnetSAModuleMSG_WithSampling, sa).forward(xyz, feat, new_xyz=torch.from_numpy(want_xyz).to(dev))
    assert torch.equal(new_feat, base_feat)
    assert cls_preds.shape == (2, 128, 3) and torch.isfinite(cls_preds).all()
    fps = copy.deepcopy(sa)
    fps.sample_type = 'D-FPS'
    with torch.no_grad():
        fx, _, _ = fps(xyz, feat)
    np.testing.assert_array_equal(fx.cpu().numpy(), np.take_along_axis(
        xyz.cpu().numpy(), oracle.furthest_point_sample(xyz.cpu().numpy(), 128)[:, :, None].astype(np.int64), 1))


def test_class_agnostic_nms(dev):
    """Restatement of model_nms_utils.py:6-28 on the HIP top-k + NMS against the same steps on the CPU oracle."""
    from oracle import cpu_oracle as oracle
    from pdm_ssd_amd.iou3d_nms import iou3d_nms_utils as iu
    rng = np.random.default_rng(5)
    n = 3000
    boxes = np.concatenate([rng.uniform(0, 40, (n, 2)), rng.uniform(-1, 1, (n, 1)), rng.uniform(1.5, 4.5, (n, 3)),
                            rng.uniform(-3.14, 3.14, (n, 1))], axis=1).astype(np.float32)
    scores = np.round(rng.uniform(0, 1, n), 2).astype(np.float32)          # rounded: plenty of score ties
    cfg = {'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.3, 'NMS_PRE_MAXSIZE': 1024, 'NMS_POST_MAXSIZE': 100}
    sel, sel_scores = iu.class_agnostic_nms(torch.from_numpy(scores).to(dev), torch.from_numpy(boxes).to(dev), cfg, score_thresh=0.2)
    mask = scores >= 0.2
    orig = np.nonzero(mask)[0]
    pre = oracle.topk_sampling(scores[mask][None], min(1024, int(mask.sum())))[0]
    keep = oracle.nms(boxes[mask][pre], 0.3)
    want = orig[pre[keep[:100]]]
    np.testing.assert_array_equal(sel.cpu().numpy(), want)
    np.testing.assert_array_equal(sel_scores.cpu().numpy(), scores[want])
