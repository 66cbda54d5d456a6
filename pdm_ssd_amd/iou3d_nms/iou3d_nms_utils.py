"""Rotated-box IoU and NMS operators on MI355X with the public names and argument meaning of
/root/reference/pcdet/ops/iou3d_nms/iou3d_nms_utils.py (boxes: (N, 7) [x, y, z, dx, dy, dz, heading]).
Every operator dispatches to a hand-written HIP kernel in libpdmssd_hip.so (include/pdmssd_hip.h, "rotated-box IoU /
NMS"); there is no CPU or PyTorch fallback.  Unlike the reference, NMS reduces its suppression mask on the device;
the only host synchronisation of `nms_gpu` is reading the number of kept boxes to size the returned tensor.
"""
import torch

from .. import _native


def _boxes(name, t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 2 and t.shape[1] == 7):
        raise ValueError(f"{name} must be a CUDA/HIP tensor of shape (N, 7)")
    return t.float().contiguous()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _pairwise(entry, boxes_a, boxes_b):
    a, b = _boxes("boxes_a", boxes_a), _boxes("boxes_b", boxes_b)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _native.call(entry, _stream(a), a.shape[0], a.data_ptr(), b.shape[0], b.data_ptr(), out.data_ptr())
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """(N, M) intersection areas of the BEV footprints (the reference's boxes_overlap_bev_gpu)."""
    return _pairwise("pdm_boxes_overlap_bev", boxes_a, boxes_b)


def boxes_iou_bev(boxes_a, boxes_b):
    """ref :34-47 — (N, M) BEV IoU."""
    return _pairwise("pdm_boxes_iou_bev", boxes_a, boxes_b)


def _iou3d_from_overlap(boxes_a, boxes_b, overlaps_bev, pairwise):
    """ref :63-79 / :103-117: height overlap x BEV overlap over the union volume."""
    shape_a, shape_b = ((-1, 1), (1, -1)) if pairwise else ((-1, 1), (-1, 1))
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(*shape_a)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(*shape_a)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(*shape_b)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(*shape_b)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(*shape_a)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(*shape_b)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """ref :50-81 — (N, M) 3-D IoU."""
    a, b = _boxes("boxes_a", boxes_a), _boxes("boxes_b", boxes_b)
    return _iou3d_from_overlap(a, b, boxes_overlap_bev(a, b), pairwise=True)


def _aligned_overlap(boxes_a, boxes_b):
    a, b = _boxes("boxes_a", boxes_a), _boxes("boxes_b", boxes_b)
    assert a.shape[0] == b.shape[0], "aligned IoU needs two box lists of the same length"
    out = torch.zeros((a.shape[0], 1), dtype=torch.float32, device=a.device)
    _native.call("pdm_boxes_aligned_overlap_bev", _stream(a), a.shape[0], a.data_ptr(), b.data_ptr(), out.data_ptr())
    return a, b, out


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b):
    """ref :83-119 — (N, 1) 3-D IoU of box i of one list with box i of the other."""
    a, b, ov = _aligned_overlap(boxes_a, boxes_b)
    return _iou3d_from_overlap(a, b, ov, pairwise=False)


def paired_boxes_iou3d_gpu(boxes_a, boxes_b):
    """ref :154-188 — (N,) 3-D IoU of paired boxes."""
    a, b, ov = _aligned_overlap(boxes_a, boxes_b)
    return _iou3d_from_overlap(a, b, ov, pairwise=False).view(-1)


def _nms(boxes, scores, thresh, pre_maxsize, normal):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    sorted_boxes = _boxes("boxes", boxes[order])
    n = sorted_boxes.shape[0]
    keep = torch.empty((n,), dtype=torch.int64, device=boxes.device)
    num = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    nbytes = _native.lib().pdm_nms_workspace_bytes(n)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=boxes.device)
    _native.call("pdm_nms", _stream(boxes), n, sorted_boxes.data_ptr(), float(thresh), 1 if normal else 0, ws.data_ptr(),
                 nbytes, keep.data_ptr(), num.data_ptr())
    return order[keep[:int(num.item())]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """ref :120-136 — rotated NMS: indices (into `boxes`) of the kept boxes in descending score order, and None."""
    return _nms(boxes, scores, thresh, pre_maxsize, normal=False)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """ref :139-152 — NMS on the axis-aligned footprints (headings ignored)."""
    return _nms(boxes, scores, thresh, None, normal=True)


def points_in_boxes_gpu(points, boxes):
    """pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:28-41 — points (B, M, 3), boxes (B, T, 7) ->
    box_idxs_of_pts (B, M) int32: the first box of the sample's list containing the point, -1 = background."""
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    B, M, _ = points.shape
    pts, bxs = points.float().contiguous(), boxes.float().contiguous()
    out = torch.full((B, M), -1, dtype=torch.int32, device=points.device)
    _native.call("pdm_points_in_boxes", _stream(points), B, boxes.shape[1], M, bxs.data_ptr(), pts.data_ptr(), out.data_ptr())
    return out


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """ref pcdet/models/model_utils/model_nms_utils.py:6-28 — score threshold, top NMS_PRE_MAXSIZE boxes by score, rotated
    NMS, first NMS_POST_MAXSIZE survivors; returns (indices into the INPUT boxes, their scores).  nms_config: an
    object or dict with NMS_TYPE ('nms_gpu' | 'nms_normal_gpu'), NMS_THRESH, NMS_PRE_MAXSIZE, NMS_POST_MAXSIZE.
    The pre-selection is pdm_topk_sampling (ties by lower index; the reference's torch.topk leaves them unspecified)."""
    from ..pointnet2_batch.pointnet2_utils import topk_sample
    get = (lambda k: nms_config[k]) if isinstance(nms_config, dict) else (lambda k: getattr(nms_config, k))
    src_box_scores = box_scores
    scores_mask = None
    if score_thresh is not None:
        scores_mask = box_scores >= score_thresh
        box_scores = box_scores[scores_mask]
        box_preds = box_preds[scores_mask]
    selected = torch.zeros((0,), dtype=torch.int64, device=box_scores.device)
    if box_scores.shape[0] > 0:
        k = min(int(get('NMS_PRE_MAXSIZE')), box_scores.shape[0])
        if k <= 16384:
            indices = topk_sample(box_scores.view(1, -1), k)[0].long()
        else:
            indices = torch.topk(box_scores, k=k)[1]
        boxes_for_nms = box_preds[indices]
        nms_fn = {'nms_gpu': nms_gpu, 'nms_normal_gpu': nms_normal_gpu}[get('NMS_TYPE')]
        keep_idx, _ = nms_fn(boxes_for_nms[:, 0:7], box_scores[indices], get('NMS_THRESH'))
        selected = indices[keep_idx[:int(get('NMS_POST_MAXSIZE'))]]
    if scores_mask is not None:
        original_idxs = scores_mask.nonzero().view(-1)
        selected = original_idxs[selected]
    return selected, src_box_scores[selected]
