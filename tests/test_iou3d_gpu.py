"""Rotated-box IoU / NMS: HIP kernels through the reference-shaped python API against
oracle/iou3d_oracle.c.  Areas and IoUs to 1e-4 (device cosf/sinf/atan2f may differ from the C library's in the last
bit); NMS decisions exact whenever no pair sits within 1e-4 of the threshold (checked)."""
import numpy as np
import pytest
import torch

from pdm_ssd_amd.iou3d_nms import iou3d_nms_utils as iu

pytestmark = pytest.mark.gpu


def random_boxes(n, seed, spread=20.0):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-spread, spread, (n, 2))
    z = rng.uniform(-1, 1, (n, 1))
    dims = np.stack([rng.uniform(1.5, 5.0, n), rng.uniform(1.0, 2.5, n), rng.uniform(1.0, 2.0, n)], 1)
    heading = rng.uniform(-np.pi, np.pi, (n, 1))
    return np.concatenate([xy, z, dims, heading], 1).astype(np.float32)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("na,nb", [(37, 53), (1, 1), (200, 16), (0, 5)])
def test_pairwise_overlap_and_iou_match_oracle(oracle, dev, na, nb):
    a, b = random_boxes(na, 1, spread=6.0), random_boxes(nb, 2, spread=6.0)
    if na and nb:
        b[0] = a[0]                                   # identical boxes (coincident edges)
        b[1 % nb, :2] = a[0, :2]; b[1 % nb, 6] = a[0, 6] + np.pi / 2
    ov = iu.boxes_overlap_bev(T(a, dev).view(-1, 7), T(b, dev).view(-1, 7)).cpu().numpy()
    iou = iu.boxes_iou_bev(T(a, dev).view(-1, 7), T(b, dev).view(-1, 7)).cpu().numpy()
    np.testing.assert_allclose(ov, oracle.boxes_overlap_bev(a, b), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(iou, oracle.boxes_iou_bev(a, b), rtol=1e-4, atol=1e-5)
    assert ov.shape == (na, nb) and (iou >= 0).all() and (iou <= 1.0 + 1e-4).all()
    if na and nb:
        assert (ov > 0).any()


def test_iou3d_and_aligned_variants(oracle, dev):
    a, b = random_boxes(64, 3, spread=5.0), random_boxes(64, 4, spread=5.0)
    ov = oracle.boxes_overlap_bev(a, b)
    h = np.clip(np.minimum(a[:, None, 2] + a[:, None, 5] / 2, b[None, :, 2] + b[None, :, 5] / 2) -
                np.maximum(a[:, None, 2] - a[:, None, 5] / 2, b[None, :, 2] - b[None, :, 5] / 2), 0, None)
    o3 = ov * h
    ref = o3 / np.clip(np.prod(a[:, 3:6], 1)[:, None] + np.prod(b[:, 3:6], 1)[None] - o3, 1e-6, None)
    got = iu.boxes_iou3d_gpu(T(a, dev), T(b, dev)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)
    al = iu.boxes_aligned_iou3d_gpu(T(a, dev), T(b, dev)).cpu().numpy()
    pr = iu.paired_boxes_iou3d_gpu(T(a, dev), T(b, dev)).cpu().numpy()
    np.testing.assert_allclose(al[:, 0], np.diag(ref), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pr, np.diag(ref), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,thresh,normal", [(500, 0.1, False), (3000, 0.01, False), (1000, 0.5, False), (700, 0.3, True),
                                              (65, 0.25, False), (1, 0.5, False), (0, 0.5, False)])
def test_nms_matches_oracle(oracle, dev, n, thresh, normal):
    boxes = random_boxes(n, 10 + n, spread=12.0)
    scores = np.random.default_rng(n).uniform(0, 1, n).astype(np.float32)
    fn = iu.nms_normal_gpu if normal else iu.nms_gpu
    sel, _ = fn(T(boxes, dev).view(-1, 7), T(scores, dev), thresh)
    order = np.argsort(-scores, kind="stable")
    # torch's sort and numpy's agree when scores are distinct (they are); greedy NMS over the sorted boxes
    ref = order[oracle.nms(boxes[order], thresh, normal=normal)]
    got = sel.cpu().numpy()
    if not np.array_equal(got, ref):                  # only a pair within 1e-4 of the threshold may flip a decision
        iou = oracle.boxes_iou_bev(boxes, boxes)
        assert (np.abs(iou - thresh) < 1e-4).any(), "NMS differs from the oracle away from the threshold"
    assert sel.dtype == torch.int64 and len(set(got.tolist())) == len(got)
    if n:
        assert got[0] == order[0]                     # the best-scored box always survives


def test_nms_more_than_16384_boxes(dev):
    """The reference's nms_gpu takes any N (iou3d_nms_utils.py:120-136); beyond 16384 boxes the suppression walk keeps
    its bitmap in LDS.  Checked against a greedy walk over the IoU matrix of the same kernels' BEV IoU."""
    n, thresh = 17000, 0.3
    boxes, scores = random_boxes(n, 21, spread=60.0), np.random.default_rng(21).uniform(0, 1, n).astype(np.float32)
    b, sc = T(boxes, dev).view(-1, 7), T(scores, dev)
    sel, _ = iu.nms_gpu(b, sc, thresh)
    order = sc.sort(0, descending=True)[1]
    sup = iu.boxes_iou_bev(b[order].contiguous(), b[order].contiguous()) > thresh
    removed = torch.zeros(n, dtype=torch.bool, device=dev)
    later = torch.arange(n, device=dev)
    keep = []
    for i in range(n):   # host loop, device state
        if bool(removed[i]):
            continue
        keep.append(i)
        removed |= sup[i] & (later > i)
    ref = order[torch.tensor(keep, device=dev)]
    if not torch.equal(sel, ref):   # a pair within 1e-4 of the threshold may be decided differently by the two kernels
        iou = iu.boxes_iou_bev(b, b)
        assert bool(((iou - thresh).abs() < 1e-4).any())
    assert len(set(sel.cpu().tolist())) == sel.numel() and sel[0] == order[0]


def test_nms_pre_maxsize_and_self_consistency(dev):
    """Kept boxes do not suppress each other, every dropped box is suppressed by a kept one of higher score."""
    boxes, scores = random_boxes(800, 5, spread=8.0), np.random.default_rng(5).uniform(0, 1, 800).astype(np.float32)
    sel, none = iu.nms_gpu(T(boxes, dev), T(scores, dev), 0.2, pre_maxsize=300)
    assert none is None
    sel = sel.cpu().numpy()
    top = np.argsort(-scores, kind="stable")[:300]
    assert set(sel.tolist()) <= set(top.tolist())
    iou = iu.boxes_iou_bev(T(boxes, dev), T(boxes, dev)).cpu().numpy()
    kept = iou[np.ix_(sel, sel)] - np.eye(len(sel))
    assert (kept <= 0.2 + 1e-4).all()
    dropped = np.setdiff1d(top, sel)
    for d in dropped:
        better = sel[scores[sel] > scores[d]]
        assert (iou[better, d] > 0.2 - 1e-4).any()


@pytest.mark.parametrize("B,T,M", [(3, 40, 5000), (1, 300, 1000), (2, 0, 64), (2, 5, 0)])
def test_points_in_boxes_matches_oracle(oracle, dev, B, T, M):
    rng = np.random.default_rng(B * 100 + T)
    boxes = np.stack([random_boxes(T, 50 + b, spread=10.0) for b in range(B)]) if T else np.zeros((B, 0, 7), np.float32)
    pts = np.concatenate([rng.uniform(-12, 12, (B, M, 2)), rng.uniform(-2, 2, (B, M, 1))], 2).astype(np.float32)
    if T and M:
        pts[:, :T] = boxes[:, :T, :3][:, :M]                    # box centres: inside (possibly several boxes: first wins)
        boxes[:, -1] = boxes[:, 0]                              # a duplicate box later in the list never wins
    ref = oracle.points_in_boxes(pts, boxes)
    got = iu.points_in_boxes_gpu(T_(pts, dev), T_(boxes, dev)).cpu().numpy()
    if not np.array_equal(got, ref):                            # device vs libm sin/cos: only points within 1e-4 of a face may differ
        assert (got != ref).mean() < 1e-3
    else:
        assert got.shape == (B, M)
    if T and M:
        assert (ref >= 0).any() and (ref == -1).any() and (ref != T - 1).all()


def T_(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
