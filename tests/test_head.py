"""Hybrid head and detector shell on the CPU, against outputs of the REFERENCE's own point head run in the authoring
container (tests/golden/ref_head.npz, generator tests/golden/gen_head_fixtures.py: PointHeadBox, PointResidualCoder,
the losses, gaussian_radius / draw_gaussian_to_heatmap of /root/reference, over an oracle-backed points_in_boxes)."""
import json
import os

import numpy as np
import pytest
import torch

from pdm_ssd_amd.config import cfg_from_dict
from pdm_ssd_amd.dense_heads import PDMHeatmapHead, PointHeadBox, point_head_template
from pdm_ssd_amd.utils import box_coder_utils, centernet_utils, loss_utils

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HEAD_CFG = {'CLS_FC': [32, 24], 'REG_FC': [24], 'CLASS_AGNOSTIC': False, 'USE_POINT_FEATURES_BEFORE_FUSION': False,
            'TARGET_CONFIG': {'GT_EXTRA_WIDTH': [0.2, 0.2, 0.2], 'BOX_CODER': 'PointResidualCoder',
                              'BOX_CODER_CONFIG': {'use_mean_size': True,
                                                   'mean_size': [[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]]}},
            'LOSS_CONFIG': {'LOSS_REG': 'WeightedSmoothL1Loss',
                            'LOSS_WEIGHTS': {'point_cls_weight': 1.0, 'point_box_weight': 2.0,
                                             'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}}}


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(G, "ref_head.npz"))


@pytest.fixture()
def cpu_points_in_boxes(oracle, monkeypatch):
    """The head asks the HIP kernel which box holds each point; on the CPU the oracle answers (test-only)."""
    def pib(points, boxes):
        return torch.from_numpy(oracle.points_in_boxes(points.detach().numpy(), boxes.detach().numpy()))
    monkeypatch.setattr(point_head_template.iou3d_nms_utils, "points_in_boxes_gpu", pib)


def make_head(ref, cfg_kind):
    cfg = cfg_from_dict(HEAD_CFG) if cfg_kind == "attr" else HEAD_CFG     # EasyDict-like or a plain dict
    head = PointHeadBox(num_class=3, input_channels=16, model_cfg=cfg)
    sd = {k[len("state."):]: torch.from_numpy(ref[k]) for k in ref.files if k.startswith("state.")}
    res = head.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return head


def test_state_dict_matches_reference_manifest(ref):
    man = json.load(open(os.path.join(G, "ref_head_manifest.json")))
    want = man['PointHeadBox(num_class=3,input_channels=16,CLS_FC=[32,24],REG_FC=[24])']
    head = PointHeadBox(num_class=3, input_channels=16, model_cfg=HEAD_CFG)
    assert {k: list(v.shape) for k, v in head.state_dict().items()} == want


@pytest.mark.parametrize("cfg_kind", ["attr", "dict"])
def test_train_forward_targets_and_losses_match_reference(ref, cpu_points_in_boxes, cfg_kind):
    head = make_head(ref, cfg_kind).train()
    bd = {'batch_size': 2, 'point_features': torch.from_numpy(ref['point_features']),
          'point_coords': torch.from_numpy(ref['point_coords']), 'gt_boxes': torch.from_numpy(ref['gt_boxes'].copy())}
    bd = head(bd)
    fr = head.forward_ret_dict
    np.testing.assert_allclose(fr['point_cls_preds'].detach().numpy(), ref['train_cls_preds'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fr['point_box_preds'].detach().numpy(), ref['train_box_preds'], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(fr['point_cls_labels'].numpy(), ref['cls_labels'])
    assert set(np.unique(ref['cls_labels']).tolist()) == {-1, 0, 1, 2, 3}      # every kind of label is exercised
    np.testing.assert_allclose(fr['point_box_labels'].numpy(), ref['box_labels'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(bd['point_cls_scores'].detach().numpy(), ref['train_scores'], rtol=1e-5, atol=1e-6)
    loss, tb = head.get_loss()
    assert abs(float(loss) - float(ref['loss'])) <= 1e-5 * abs(float(ref['loss']))
    assert abs(float(tb['point_loss_cls']) - float(ref['loss_cls'])) <= 1e-5 * abs(float(ref['loss_cls']))
    assert abs(float(tb['point_loss_box']) - float(ref['loss_box'])) <= 1e-5 * abs(float(ref['loss_box']))
    assert float(tb['point_pos_num']) == float(ref['pos_num'])
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in head.parameters())


def test_ragged_samples_take_the_per_sample_path(ref, cpu_points_in_boxes):
    """Dropping points of the second sample only: labels of the remaining points are unchanged."""
    head = make_head(ref, "attr").train()
    keep = np.ones(len(ref['point_coords']), dtype=bool)
    keep[-7:] = False
    bd = {'batch_size': 2, 'point_features': torch.from_numpy(ref['point_features'][keep]),
          'point_coords': torch.from_numpy(ref['point_coords'][keep]), 'gt_boxes': torch.from_numpy(ref['gt_boxes'].copy())}
    head(bd)
    np.testing.assert_array_equal(head.forward_ret_dict['point_cls_labels'].numpy(), ref['cls_labels'][keep])
    np.testing.assert_allclose(head.forward_ret_dict['point_box_labels'].numpy(), ref['box_labels'][keep], rtol=1e-6, atol=1e-6)


def test_class_outside_the_coder_table_poisons_the_box_targets(ref, cpu_points_in_boxes):
    """The reference asserts on a foreground class beyond the mean-size table (box_coder_utils.py:166, a host read-back
    per step); here the step stays free of synchronisations and the box targets — hence the loss — turn NaN."""
    head = make_head(ref, "attr").train()
    gt = ref['gt_boxes'].copy()
    fg_class = int(ref['cls_labels'][ref['cls_labels'] > 0][0])
    gt[..., 7][gt[..., 7] == fg_class] = 7          # a class of boxes that DO hold points
    bd = {'batch_size': 2, 'point_features': torch.from_numpy(ref['point_features']),
          'point_coords': torch.from_numpy(ref['point_coords']), 'gt_boxes': torch.from_numpy(gt)}
    head(bd)
    assert torch.isnan(head.forward_ret_dict['point_box_labels']).all()
    # ... and the LOSS: WeightedSmoothL1Loss treats a NaN target as "ignore", so the flag must reach the loss itself
    loss, tb = head.get_loss()
    assert not torch.isfinite(loss), "a foreground class outside the mean-size table must not train on silently"
    assert not torch.isfinite(tb['point_loss_box']) and torch.isfinite(tb['point_loss_cls'])
    # in-range classes: finite as before
    head2 = make_head(ref, "attr").train()
    bd2 = dict(bd, gt_boxes=torch.from_numpy(ref['gt_boxes']))
    head2(bd2)
    assert bool(head2.forward_ret_dict['point_box_labels_ok']) and torch.isfinite(head2.get_loss()[0])
    # the direct call keeps the reference's assert
    coder = box_coder_utils.PointResidualCoder(code_size=8, use_mean_size=True, mean_size=[[3.9, 1.6, 1.56], [0.8, 0.6, 1.73]])
    with pytest.raises(AssertionError):
        coder.encode_torch(torch.rand(4, 7) + 0.5, torch.rand(4, 3), torch.tensor([1, 2, 3, 1]))


def test_eval_forward_decodes_boxes_like_the_reference(ref):
    head = make_head(ref, "attr").eval()
    with torch.no_grad():
        bd = head({'batch_size': 2, 'point_features': torch.from_numpy(ref['point_features']),
                   'point_coords': torch.from_numpy(ref['point_coords'])})
    np.testing.assert_allclose(bd['batch_cls_preds'].numpy(), ref['eval_cls_preds'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bd['batch_box_preds'].numpy(), ref['eval_box_preds'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(bd['point_cls_scores'].numpy(), ref['eval_scores'], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(bd['batch_index'].numpy(), ref['eval_batch_index'])
    assert bd['cls_preds_normalized'] is False


def test_point_residual_coder(ref):
    coder = box_coder_utils.PointResidualCoder(code_size=8, use_mean_size=True,
                                               mean_size=HEAD_CFG['TARGET_CONFIG']['BOX_CODER_CONFIG']['mean_size'])
    boxes, cls, pts = (torch.from_numpy(ref[k]) for k in ('coder_boxes', 'coder_cls', 'coder_points'))
    code = coder.encode_torch(boxes.clone(), pts, cls)
    np.testing.assert_allclose(code.numpy(), ref['coder_code'], rtol=1e-6, atol=1e-6)
    dec = coder.decode_torch(code, pts, cls)
    np.testing.assert_allclose(dec.numpy(), ref['coder_decoded'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dec.numpy(), ref['coder_boxes'], rtol=1e-4, atol=1e-5)      # round trip


def test_gaussian_targets_and_losses(ref):
    hw = torch.from_numpy(ref['radius_in'])
    np.testing.assert_allclose(centernet_utils.gaussian_radius(hw[:, 0], hw[:, 1], min_overlap=0.1).numpy(), ref['radius_out'],
                               rtol=1e-6)
    # batched device-side drawing == the reference's box-by-box drawing
    c, r = torch.from_numpy(ref['draw_centers']), torch.from_numpy(ref['draw_radius'])
    hm = torch.zeros(1, 1, 24, 30)
    cls_idx = torch.zeros(1, len(c), 2, dtype=torch.long)
    centernet_utils.draw_gaussians(hm, cls_idx, c[None], r[None], torch.ones(1, len(c), dtype=torch.bool))
    np.testing.assert_allclose(hm[0, 0].numpy(), ref['draw_heatmap'], rtol=1e-6, atol=1e-7)
    pred, tgt = torch.from_numpy(ref['focal_pred']), torch.from_numpy(ref['focal_target'])
    fl = loss_utils.FocalLossCenterNet()
    assert abs(float(fl(pred, tgt)) - float(ref['focal_loss'])) <= 1e-5 * abs(float(ref['focal_loss']))
    assert abs(float(fl(pred, torch.zeros_like(tgt))) - float(ref['focal_loss_empty'])) <= 1e-5 * abs(float(ref['focal_loss_empty']))
    sf = loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)
    got = sf(torch.from_numpy(ref['sfl_x']), torch.from_numpy(ref['sfl_t']), torch.from_numpy(ref['sfl_w']))
    np.testing.assert_allclose(got.numpy(), ref['sfl_out'], rtol=1e-5, atol=1e-7)
    # mixed dtypes promote as the reference's arithmetic form does (fp32 one-hot targets with bf16 logits)
    xb = torch.from_numpy(ref['sfl_x']).to(torch.bfloat16)
    mixed = sf(xb, torch.from_numpy(ref['sfl_t']), torch.from_numpy(ref['sfl_w']))
    assert mixed.dtype == torch.float32
    np.testing.assert_allclose(mixed.numpy(), sf(xb.float(), torch.from_numpy(ref['sfl_t']), torch.from_numpy(ref['sfl_w'])).numpy(), rtol=1e-6)


def test_heatmap_head_targets_and_loss():
    """Build-defined dense half of the hybrid head: gaussian target at the cell of every box centre, focal loss finite
    and lower for a prediction that matches the target."""
    cfg = {'NAME': 'PDMHeatmapHead', 'SHARED_CONV_CHANNEL': 8, 'NUM_CONTEXT_CONV': 1,
           'TARGET_ASSIGNER_CONFIG': {'FEATURE_MAP_STRIDE': 8, 'GAUSSIAN_OVERLAP': 0.1, 'MIN_RADIUS': 2},
           'LOSS_CONFIG': {'LOSS_WEIGHTS': {'cls_weight': 1.0}}}
    head = PDMHeatmapHead(cfg, input_channels=4, num_class=3, point_cloud_range=[0, -40, -3, 70.4, 40, 1],
                          voxel_size=[0.05, 0.05, 0.1]).train()
    gt = torch.zeros(2, 3, 8)
    gt[0, 0] = torch.tensor([10.0, 0.0, -1, 3.9, 1.6, 1.5, 0.3, 1])
    gt[0, 1] = torch.tensor([30.2, -20.0, -1, 0.8, 0.6, 1.7, 0.0, 2])
    gt[1, 0] = torch.tensor([69.9, 39.9, -1, 1.7, 0.6, 1.7, 1.0, 3])          # last cell of the map
    x = torch.randn(2, 4, 200, 176)
    out = head({'spatial_features': x, 'gt_boxes': gt})
    hm = head.forward_ret_dict['heatmap']
    assert tuple(hm.shape) == (2, 3, 200, 176) and tuple(out['bev_heatmap'].shape) == (2, 3, 200, 176)
    assert hm[0, 0, 100, 25] == 1.0 and hm[0, 1, 50, 75] == 1.0 and hm[1, 2, 199, 174] == 1.0     # (y, x) cells of the centres
    assert int((hm == 1.0).sum()) == 3 and float(hm[1, 0].abs().sum()) == 0.0                       # padding rows draw nothing
    assert 0 < float(hm[0, 0, 100, 26]) < 1
    loss, tb = head.get_loss()
    assert torch.isfinite(loss) and 'hm_loss' in tb
    head.forward_ret_dict['hm_logits'] = torch.logit(hm.clamp(1e-4, 1 - 1e-4))
    assert float(head.get_loss()[0]) < float(loss)


def test_detector_registry_and_module_slots():
    """Detector assembly by NAME (detector3d_template.py:68-139): slots, constructor keywords, channel threading."""
    from pdm_ssd_amd import detectors
    from pdm_ssd_amd.detector_config import PDM_SSD_CFG, build_pdm_ssd
    assert set(detectors.__all__) >= {'PDMSSD', 'Detector3DTemplate'}
    assert 'PointNet2MSG' in detectors.BACKBONES_3D and 'PDMNeck' in detectors.MAP_TO_BEV
    model = build_pdm_ssd()
    assert [type(m).__name__ for m in model.module_list] == ['PointNet2MSG', 'PDMNeck', 'PDMHeatmapHead', 'PointHeadBox']
    assert model.backbone_3d.num_point_features == 128 and model.map_to_bev_module.num_bev_features == 128
    assert model.point_head.cls_layers[0].in_features == 128 and model.dense_head.shared_conv[0].in_channels == 128
    assert model.dense_head.shared_conv[0].groups == 128 and model.dense_head.shared_conv[3].out_channels == 64   # separable block
    assert model.vfe is None and model.roi_head is None and int(model.global_step) == 0
    keys = set(model.state_dict())
    assert {'backbone_3d.SA_modules.0.mlps.0.0.weight', 'map_to_bev_module.proj.0.weight', 'dense_head.hm.2.bias',
            'point_head.box_layers.6.bias', 'global_step'} <= keys
    bad = dict(PDM_SSD_CFG, VFE={'NAME': 'MeanVFE'})
    with pytest.raises(AssertionError, match="spconv"):
        build_pdm_ssd(bad)
    missing = model.load_params_from_state_dict({k: v for k, v in model.state_dict().items() if 'point_head' not in k}, strict=False)
    assert missing and all('point_head' in k for k in missing)
    # strict (the default, ref detector3d_template.py:354-358): a checkpoint that lacks parameters of the model raises,
    # one whose entry has another shape raises too (the entry is dropped, hence missing); a complete one loads
    partial = {k: v for k, v in model.state_dict().items() if 'point_head' not in k}
    with pytest.raises(RuntimeError, match="Missing key"):
        model.load_params_from_state_dict(partial)
    reshaped = dict(model.state_dict())
    reshaped['point_head.box_layers.6.bias'] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="Missing key"):
        model.load_params_from_state_dict(reshaped)
    full = {k: v.clone() + (1.0 if v.is_floating_point() else 0) for k, v in model.state_dict().items()}
    assert model.load_params_from_state_dict(full) == []
    assert torch.equal(model.state_dict()['point_head.box_layers.6.bias'], full['point_head.box_layers.6.bias'])
