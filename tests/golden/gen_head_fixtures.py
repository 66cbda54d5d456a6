#!/usr/bin/env python3
"""Generates tests/golden/ref_head.npz by running the REFERENCE's own point head on the CPU:
/root/reference/pcdet/models/dense_heads/point_head_box.py + point_head_template.py, utils/box_coder_utils.py,
utils/loss_utils.py, utils/box_utils.py, models/model_utils/centernet_utils.py — imported from where they lie,
nothing copied — with only what this image lacks replaced:
  - the native `points_in_boxes_gpu` (roiaware_pool3d_cuda) by a stub over this repo's CPU oracle,
  - `.cuda()` by the identity (the reference's coders / losses move constants to the GPU in their constructors),
  - SharedArray / numba / iou3d_nms_utils / spconv-dependent packages by empty modules (imported, never called here).
Records: state_dict manifest, forward outputs (logits, codes, decoded boxes), target labels, the three losses, the
coder round trip, gaussian_radius / draw_gaussian_to_heatmap and FocalLossCenterNet values for seeded inputs.
Run in the authoring container only (needs /root/reference); the .npz / .json outputs are committed.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import cpu_oracle as o  # noqa: E402

REF = '/root/reference'


class EasyDict(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = EasyDict(v) if isinstance(v, dict) else v

    __getattr__ = dict.__getitem__


def install_reference():
    def pkg(name, path=None):
        m = types.ModuleType(name)
        m.__path__ = [path] if path else []
        sys.modules[name] = m
        return m
    pkg('pcdet', f'{REF}/pcdet')
    pkg('pcdet.ops', f'{REF}/pcdet/ops')
    pkg('pcdet.utils', f'{REF}/pcdet/utils')
    pkg('pcdet.models', f'{REF}/pcdet/models')
    pkg('pcdet.models.dense_heads', f'{REF}/pcdet/models/dense_heads')     # package __init__ not run (it imports every head)
    pkg('pcdet.models.model_utils', f'{REF}/pcdet/models/model_utils')
    sys.modules['SharedArray'] = types.ModuleType('SharedArray')
    numba = types.ModuleType('numba')
    numba.jit = lambda *a, **k: (lambda f: f)
    sys.modules['numba'] = numba
    iou = pkg('pcdet.ops.iou3d_nms')
    iou.iou3d_nms_utils = types.ModuleType('pcdet.ops.iou3d_nms.iou3d_nms_utils')
    sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_utils'] = iou.iou3d_nms_utils
    roi = pkg('pcdet.ops.roiaware_pool3d')
    ru = types.ModuleType('pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils')

    def points_in_boxes_gpu(points, boxes):
        return torch.from_numpy(o.points_in_boxes(points.detach().numpy(), boxes.detach().numpy()))
    ru.points_in_boxes_gpu = points_in_boxes_gpu
    roi.roiaware_pool3d_utils = ru
    sys.modules[ru.__name__] = ru
    torch.Tensor.cuda = lambda self, *a, **k: self
    from pcdet.models.dense_heads import point_head_box          # the reference's files
    from pcdet.models.model_utils import centernet_utils
    from pcdet.utils import box_coder_utils, loss_utils
    return point_head_box, centernet_utils, box_coder_utils, loss_utils


HEAD_CFG = {'CLS_FC': [32, 24], 'REG_FC': [24], 'CLASS_AGNOSTIC': False, 'USE_POINT_FEATURES_BEFORE_FUSION': False,
            'TARGET_CONFIG': {'GT_EXTRA_WIDTH': [0.2, 0.2, 0.2], 'BOX_CODER': 'PointResidualCoder',
                              'BOX_CODER_CONFIG': {'use_mean_size': True,
                                                   'mean_size': [[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]]}},
            'LOSS_CONFIG': {'LOSS_REG': 'WeightedSmoothL1Loss',
                            'LOSS_WEIGHTS': {'point_cls_weight': 1.0, 'point_box_weight': 2.0,
                                             'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}}}


def scene(rng, B, n, M):
    """n points per sample in a 40 x 40 x 4 m block, M boxes of the three KITTI classes (last rows zero = padding)."""
    gt = np.zeros((B, M, 8), dtype=np.float32)
    pts = []
    for b in range(B):
        k = M - b            # sample b carries b padding rows
        gt[b, :k, 0:2] = rng.uniform(5, 35, (k, 2))
        gt[b, :k, 2] = rng.uniform(-1.2, -0.6, k)
        cls = rng.integers(1, 4, k)
        sizes = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], dtype=np.float32)[cls - 1]
        gt[b, :k, 3:6] = sizes * rng.uniform(0.8, 1.2, (k, 3))
        gt[b, :k, 6] = rng.uniform(-np.pi, np.pi, k)
        gt[b, :k, 7] = cls
        p = np.stack([rng.uniform(0, 40, n), rng.uniform(0, 40, n), rng.uniform(-3, 1, n)], 1)
        near = rng.integers(0, k, n // 2)   # half of the points near some box so every label kind occurs
        p[:n // 2] = gt[b, near, 0:3] + rng.normal(0, 1.0, (n // 2, 3)) * [1.2, 0.6, 0.5]
        pts.append(np.concatenate([np.full((n, 1), b), p], 1))
    return np.concatenate(pts).astype(np.float32), gt


def main():
    phb, cu, bcu, lu = install_reference()
    rng = np.random.default_rng(77)
    out, manifest = {}, {}
    torch.manual_seed(5)
    head = phb.PointHeadBox(num_class=3, input_channels=16, model_cfg=EasyDict(HEAD_CFG))
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for name, buf in head.named_buffers():
            if name.endswith('running_mean'):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
            elif name.endswith('running_var'):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
    manifest['PointHeadBox(num_class=3,input_channels=16,CLS_FC=[32,24],REG_FC=[24])'] = \
        {k: list(v.shape) for k, v in head.state_dict().items()}
    for k, v in head.state_dict().items():
        out['state.' + k] = v.numpy()
    B, n, M = 2, 300, 6
    coords, gt = scene(rng, B, n, M)
    feats = rng.standard_normal((B * n, 16)).astype(np.float32)
    out.update(point_coords=coords, gt_boxes=gt, point_features=feats)

    head.train()
    bd = {'batch_size': B, 'point_features': torch.from_numpy(feats), 'point_coords': torch.from_numpy(coords),
          'gt_boxes': torch.from_numpy(gt.copy())}
    bd = head(bd)
    loss, tb = head.get_loss()
    fr = head.forward_ret_dict
    out.update(train_cls_preds=fr['point_cls_preds'].detach().numpy(), train_box_preds=fr['point_box_preds'].detach().numpy(),
               cls_labels=fr['point_cls_labels'].numpy(), box_labels=fr['point_box_labels'].numpy(),
               loss=np.float32(loss.item()), loss_cls=np.float32(tb['point_loss_cls']), loss_box=np.float32(tb['point_loss_box']),
               pos_num=np.float32(tb['point_pos_num']), train_scores=bd['point_cls_scores'].detach().numpy())
    head.eval()
    with torch.no_grad():
        bd = head({'batch_size': B, 'point_features': torch.from_numpy(feats), 'point_coords': torch.from_numpy(coords)})
    out.update(eval_cls_preds=bd['batch_cls_preds'].numpy(), eval_box_preds=bd['batch_box_preds'].numpy(),
               eval_scores=bd['point_cls_scores'].numpy(), eval_batch_index=bd['batch_index'].numpy())

    # coder round trip and raw values
    coder = bcu.PointResidualCoder(code_size=8, use_mean_size=True, mean_size=HEAD_CFG['TARGET_CONFIG']['BOX_CODER_CONFIG']['mean_size'])
    boxes = torch.from_numpy(gt[0, :5, :7].copy())
    cls = torch.from_numpy(gt[0, :5, 7].astype(np.int64))
    pts = boxes[:, :3] + torch.tensor([[0.3, -0.2, 0.1]])
    code = coder.encode_torch(boxes.clone(), pts, cls)
    out.update(coder_boxes=boxes.numpy(), coder_cls=cls.numpy(), coder_points=pts.numpy(), coder_code=code.numpy(),
               coder_decoded=coder.decode_torch(code, pts, cls).numpy())

    # heat-map pieces
    hw = torch.from_numpy(rng.uniform(0.5, 12, (20, 2)).astype(np.float32))
    out.update(radius_in=hw.numpy(), radius_out=cu.gaussian_radius(hw[:, 0], hw[:, 1], min_overlap=0.1).numpy())
    hm = torch.zeros(24, 30)
    draws = [((5, 7), 2), ((28, 22), 3), ((0, 0), 2), ((6, 8), 4), ((15, 12), 1)]
    for c, r in draws:
        cu.draw_gaussian_to_heatmap(hm, torch.tensor(c, dtype=torch.float32), r)
    out.update(draw_centers=np.array([c for c, _ in draws], dtype=np.int64), draw_radius=np.array([r for _, r in draws], dtype=np.int64),
               draw_heatmap=hm.numpy())
    pred = torch.from_numpy(rng.uniform(0.01, 0.99, (2, 3, 24, 30)).astype(np.float32))
    tgt = torch.zeros(2, 3, 24, 30)
    tgt[0, 0] = hm
    tgt[1, 2] = hm.flip(0)
    out.update(focal_pred=pred.numpy(), focal_target=tgt.numpy(),
               focal_loss=np.float32(lu.FocalLossCenterNet()(pred, tgt).item()),
               focal_loss_empty=np.float32(lu.FocalLossCenterNet()(pred, torch.zeros_like(tgt)).item()))
    sf = lu.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)
    x = torch.from_numpy(rng.standard_normal((1, 50, 3)).astype(np.float32))
    t = torch.zeros(1, 50, 3); t[0, torch.arange(50), torch.from_numpy(rng.integers(0, 3, 50))] = 1
    w = torch.from_numpy(rng.uniform(0, 1, (1, 50)).astype(np.float32))
    out.update(sfl_x=x.numpy(), sfl_t=t.numpy(), sfl_w=w.numpy(), sfl_out=sf(x, t, w).numpy())

    np.savez_compressed(os.path.join(HERE, 'ref_head.npz'), **out)
    with open(os.path.join(HERE, 'ref_head_manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print('wrote', len(out), 'arrays; labels:', {int(v): int((out['cls_labels'] == v).sum()) for v in np.unique(out['cls_labels'])},
          'loss', float(loss))


if __name__ == '__main__':
    main()
