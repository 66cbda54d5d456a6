"""Hybrid head + detector shell on the GPU: the point head's HIP paths (points_in_boxes target assignment, fused per-row
MLP kernels) against the reference-run fixtures / the torch layers, and the detector's two contracts
(/root/reference/pcdet/models/detectors/point_rcnn.py:9-24): training returns ({'loss'}, tb_dict, disp_dict), eval
returns per-sample prediction dicts after NMS."""
import copy
import os

import numpy as np
import pytest
import torch

from pdm_ssd_amd import detectors, synthetic
from pdm_ssd_amd.dense_heads import PointHeadBox
from pdm_ssd_amd.detector_config import PDM_SSD_CFG, build_pdm_ssd

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SMALL = dict(PDM_SSD_CFG)
SMALL['BACKBONE_3D'] = {'NAME': 'PointNet2MSG',
                        'SA_CONFIG': {'NPOINTS': [512, 128, 32], 'RADIUS': [[0.5, 1.0], [1.0, 2.0], [2.0, 4.0]],
                                      'NSAMPLE': [[16, 32], [16, 32], [16, 32]],
                                      'MLPS': [[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]],
                                               [[128, 196, 256], [128, 196, 256]]]},
                        'FP_MLPS': [[128, 128], [256, 256], [512, 512]]}
SMALL['MAP_TO_BEV'] = dict(PDM_SSD_CFG['MAP_TO_BEV'], FEATURE_DIM=32, DILATION=[5, 5, 1])


HEAD_CFG = {'CLS_FC': [32, 24], 'REG_FC': [24], 'CLASS_AGNOSTIC': False, 'USE_POINT_FEATURES_BEFORE_FUSION': False,
            'TARGET_CONFIG': {'GT_EXTRA_WIDTH': [0.2, 0.2, 0.2], 'BOX_CODER': 'PointResidualCoder',
                              'BOX_CODER_CONFIG': {'use_mean_size': True,
                                                   'mean_size': [[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]]}},
            'LOSS_CONFIG': {'LOSS_REG': 'WeightedSmoothL1Loss',
                            'LOSS_WEIGHTS': {'point_cls_weight': 1.0, 'point_box_weight': 2.0,
                                             'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}}}   # as tests/golden/gen_head_fixtures.py


def scene_boxes(B, M, seed):
    rng = np.random.default_rng(seed)
    gt = np.zeros((B, M, 8), dtype=np.float32)
    sizes = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], dtype=np.float32)
    for b in range(B):
        k = M - b
        cls = rng.integers(1, 4, k)
        gt[b, :k, 0] = rng.uniform(5, 60, k); gt[b, :k, 1] = rng.uniform(-30, 30, k); gt[b, :k, 2] = rng.uniform(-1.5, -0.5, k)
        gt[b, :k, 3:6] = sizes[cls - 1] * rng.uniform(0.9, 1.1, (k, 3))
        gt[b, :k, 6] = rng.uniform(-np.pi, np.pi, k)
        gt[b, :k, 7] = cls
    return gt


@pytest.mark.gpu
def test_point_head_targets_on_gpu_match_reference_fixture(dev):
    """points_in_boxes HIP kernel + batched target assignment == the reference's labels (oracle-backed fixture)."""
    ref = np.load(os.path.join(G, "ref_head.npz"))
    head = PointHeadBox(num_class=3, input_channels=16, model_cfg=HEAD_CFG)
    head.load_state_dict({k[len("state."):]: torch.from_numpy(ref[k]) for k in ref.files if k.startswith("state.")})
    head = head.to(dev).train()
    bd = {'batch_size': 2, 'point_features': torch.from_numpy(ref['point_features']).to(dev),
          'point_coords': torch.from_numpy(ref['point_coords']).to(dev), 'gt_boxes': torch.from_numpy(ref['gt_boxes'].copy()).to(dev)}
    head(bd)
    np.testing.assert_array_equal(head.forward_ret_dict['point_cls_labels'].cpu().numpy(), ref['cls_labels'])
    np.testing.assert_allclose(head.forward_ret_dict['point_box_labels'].cpu().numpy(), ref['box_labels'], rtol=1e-5, atol=1e-5)
    loss, tb = head.get_loss()
    assert abs(float(loss) - float(ref['loss'])) <= 1e-4 * abs(float(ref['loss']))


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True])
def test_point_head_fused_loss_equals_torch_formulation(dev, bf16):
    """pdm_point_head_loss (targets + focal + smooth-L1 + gradients in three launches) against the torch formulation of
    point_head_template.py (the one pinned by the reference-run fixtures): labels identical, both losses and the positive
    count to 1e-5, the gradients of every parameter to 1e-4 of their scale — fp32, and bf16 predictions under autocast (the
    gradient of a bf16 prediction is rounded once on both paths).  Zero-padded box rows, points inside only the enlarged box
    and a sample without a positive are in the case."""
    torch.manual_seed(4)
    B, n = 3, 2048
    gt = scene_boxes(B, 5, 11)
    gt[2] = 0.0                                            # a sample with no box at all
    rng = np.random.default_rng(5)
    xyz = np.stack([rng.uniform(0, 70, (B, n)), rng.uniform(-40, 40, (B, n)), rng.uniform(-3, 1, (B, n))], -1).astype(np.float32)
    for b in range(2):
        for k in range(3):     # points in and just around the first boxes (inside / ring of the enlarged box / outside)
            xyz[b, 300 * k:300 * (k + 1)] = gt[b, k, :3] + rng.normal(0, 1.0, (300, 3)).astype(np.float32) * gt[b, k, 3:6] * 0.4
    coords = np.concatenate([np.repeat(np.arange(B, dtype=np.float32), n)[:, None], xyz.reshape(-1, 3)], 1)
    head = PointHeadBox(num_class=3, input_channels=16, model_cfg=HEAD_CFG).to(dev).train()
    feats = torch.randn(B * n, 16, device=dev)
    bd = {'batch_size': B, 'point_features': feats, 'point_coords': torch.from_numpy(coords).to(dev),
          'gt_boxes': torch.from_numpy(gt).to(dev), 'points_per_sample_checked': True}
    res = {}
    for fused_on in (True, False):
        head.use_fused_loss = fused_on
        head.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            head(dict(bd))
            assert ('fused_loss_inputs' in head.forward_ret_dict) == fused_on
            loss, tb = head.get_loss()
        loss.backward()
        res[fused_on] = (float(loss), {k: float(v) for k, v in tb.items()}, head.forward_ret_dict['point_cls_labels'].clone(),
                         {k: p.grad.clone() for k, p in head.named_parameters()})
    (la, ta, laba, ga), (lb, tb_, labb, gb) = res[True], res[False]
    assert torch.equal(laba, labb)
    assert int((laba > 0).sum()) > 50 and int((laba == -1).sum()) > 5 and ta['point_pos_num'] == tb_['point_pos_num'] == float((laba > 0).sum())
    tol = 2e-3 if bf16 else 1e-5
    assert abs(la - lb) <= tol * abs(lb)
    for k in ('point_loss_cls', 'point_loss_box'):
        assert abs(ta[k] - tb_[k]) <= tol * abs(tb_[k]) + 1e-7, k
    for k in ga:
        scale = float(gb[k].abs().max())
        assert float((ga[k] - gb[k]).abs().max()) <= (2e-2 if bf16 else 1e-4) * scale + 1e-9, k
    # a positive whose class is outside the mean-size table: NaN box loss (the reference asserts)
    head.use_fused_loss = True
    bad = dict(bd, gt_boxes=bd['gt_boxes'].clone())
    bad['gt_boxes'][0, 0, 7] = 9
    head(bad)
    loss, tb = head.get_loss()
    assert not torch.isfinite(loss) and not torch.isfinite(tb['point_loss_box']) and torch.isfinite(tb['point_loss_cls'])


@pytest.mark.gpu
def test_point_head_fused_inference_equals_torch_layers(dev):
    torch.manual_seed(0)
    head = build_pdm_ssd().point_head.to(dev).eval()
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    n = 2 * 4096
    bd = {'batch_size': 2, 'point_features': torch.randn(n, 128, device=dev),
          'point_coords': torch.cat([torch.arange(2, device=dev).repeat_interleave(n // 2)[:, None].float(),
                                     torch.rand(n, 3, device=dev) * 40], dim=1)}
    with torch.no_grad():
        a = head(dict(bd))
        assert head._pdm_fused_cache['cls'][1] is not None          # the MFMA row kernels really ran
        head.use_fused = False
        b = head(dict(bd))
    for k in ('batch_cls_preds', 'batch_box_preds', 'point_cls_scores'):
        torch.testing.assert_close(a[k], b[k], rtol=1e-4, atol=1e-4)
    assert tuple(a['batch_box_preds'].shape) == (n, 7) and tuple(a['batch_cls_preds'].shape) == (n, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8192, 8192 + 37, 3 * 16384 + 1])
def test_point_head_one_launch_equals_two_launches(dev, n):
    """The class and box stacks of the point head as ONE launch (pdm_rows_mlp_fused_pair: a wave keeps its 16 rows' input
    fragments and runs both chains, six layers of one continuous weight stream) against the two-launch form: the same
    per-layer arithmetic, so logits, box codes, decoded boxes and scores are BIT-identical — ragged last tile and several
    tiles per workgroup included."""
    from pdm_ssd_amd import _native
    torch.manual_seed(3)
    head = build_pdm_ssd().point_head.to(dev).eval()
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    bd = {'batch_size': 1, 'point_features': torch.randn(n, 128, device=dev),
          'point_coords': torch.cat([torch.zeros(n, 1, device=dev), torch.rand(n, 3, device=dev) * 40], dim=1)}
    calls = []
    orig = _native.call
    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    _native.call = spy
    cap = _native.lib().pdm_tune_rows_chain_wg_per_cu(1 if n > 16384 else 12)   # 256 workgroups for 769 tiles: several tiles each
    try:
        with torch.no_grad():
            one = head(dict(bd))
            first = list(calls); calls.clear()
            head.use_pair = False
            two = head(dict(bd))
            second = list(calls); calls.clear()
            old = _native.lib().pdm_tune_fused_pair(0)          # the entry point's own two-launch route
            head.use_pair = True
            three = head(dict(bd))
            _native.lib().pdm_tune_fused_pair(old)
    finally:
        _native.call = orig
        _native.lib().pdm_tune_rows_chain_wg_per_cu(cap)
    assert first.count("pdm_rows_mlp_fused_pair") == 1 and first.count("pdm_rows_mlp_fused") == 0
    assert second.count("pdm_rows_mlp_fused") == 2 and second.count("pdm_rows_mlp_fused_pair") == 0
    for k in ('batch_cls_preds', 'batch_box_preds', 'point_cls_scores'):
        assert torch.isfinite(one[k]).all()
        assert torch.equal(one[k], two[k]), k
        assert torch.equal(one[k], three[k]), k
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,kind", [(2, 16384, "uniform"), (3, 16384, "lidar")])
def test_last_fp_module_inside_the_point_head_launch_equals_separate_launches(dev, B, N, kind):
    """pdm_fp_head_fused (the backbone's last FP module + both stacks of the point head, one launch) against the separate
    launches: bit-equal to FP-through-the-chain-kernel + pdm_rows_mlp_fused_pair, 1e-4 against the LDS-tiled FP kernel
    (another summation order), and the module's rows (point_features) are really written."""
    from pdm_ssd_amd import _native
    torch.manual_seed(5)
    model = build_pdm_ssd().to(dev).eval()
    cl = synthetic.uniform_clouds(B, N, 3) if kind == "uniform" else synthetic.lidar_like_clouds(B, N, 3)
    pts = torch.from_numpy(synthetic.to_batch_points(cl)).to(dev)
    calls = []
    orig = _native.call
    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    keys = ('point_features', 'batch_cls_preds', 'batch_box_preds', 'point_cls_scores')
    def run(fusion, mask):
        model.point_head.use_fp_fusion = fusion
        old = _native.lib().pdm_tune_fp_chain_mask(mask)
        calls.clear()
        _native.call = spy
        try:
            with torch.no_grad():
                bd = {'batch_size': B, 'points': pts.clone(), 'points_per_sample_checked': True}
                if model.point_head.wants_deferred_fp():
                    bd['defer_last_fp'] = True
                bd = model.point_head(model.backbone_3d(bd))
                assert 'point_features_deferred' not in bd
        finally:
            _native.call = orig
            _native.lib().pdm_tune_fp_chain_mask(old)
        return {k: bd[k].clone() for k in keys}, list(calls)
    fusedv, c1 = run(True, 3)
    apart, c2 = run(False, 3)          # FP1 through the chain kernel, then the pair launch
    tiled, c3 = run(False, 2)          # the default of the unfused path: FP1 through the LDS-tiled kernel
    model.point_head.use_fp_fusion = False      # (the default: an opt-in form)
    assert c1.count("pdm_fp_head_fused") == 1 and c1.count("pdm_rows_mlp_fused_pair") == 0
    assert c2.count("pdm_fp_head_fused") == 0 and c2.count("pdm_rows_mlp_fused_pair") == 1
    assert c1.count("pdm_fp_mlp_fused_pre") == c2.count("pdm_fp_mlp_fused_pre") - 1       # the last FP module went into the head's launch
    for k in keys:
        assert torch.isfinite(fusedv[k]).all(), k
        assert torch.equal(fusedv[k], apart[k]), k
        scale = float(tiled[k].abs().max()) + 1e-6
        assert float((fusedv[k] - tiled[k]).abs().max()) <= 1e-4 * max(1.0, scale), k
    assert float(fusedv['point_features'].abs().max()) > 0


@pytest.mark.gpu
def test_detector_training_contract_and_backward(dev):
    torch.manual_seed(1)
    model = build_pdm_ssd(SMALL).to(dev).train()
    B, N = 2, 2048
    cl = synthetic.lidar_like_clouds(B, N, 5)
    gt = scene_boxes(B, 6, 3)
    cl[:, :200, :3] = gt[:, :1, :3] + np.random.default_rng(0).normal(0, 0.5, (B, 200, 3)).astype(np.float32)   # points on a box
    batch = {'batch_size': B, 'points': torch.from_numpy(synthetic.to_batch_points(cl)).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev)}
    ret = detectors.model_fn_decorator()(model, batch)
    assert set(ret._fields) == {'loss', 'tb_dict', 'disp_dict'}
    assert ret.loss.dim() == 0 and torch.isfinite(ret.loss) and int(model.global_step) == 1
    assert {'point_loss_cls', 'point_loss_box', 'point_pos_num', 'hm_loss'} <= set(ret.tb_dict)
    assert float(ret.tb_dict['point_pos_num']) > 0
    ret.loss.backward()
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name


@pytest.mark.gpu
def test_training_step_issues_without_host_synchronisation(dev):
    """Forward + losses of a training step contain no blocking call (torch's sync debug mode raises at a size read back
    for a boolean-mask index, an .item(), a pageable host->device copy ...): the host can issue the backward ahead of
    the device.  (bench.py --train-host-profile lists them for the bench model; the reference has three per step.)"""
    torch.manual_seed(1)
    model = build_pdm_ssd(SMALL).to(dev).train()
    B, N = 2, 2048
    cl = synthetic.lidar_like_clouds(B, N, 5)
    gt = scene_boxes(B, 6, 3)
    batch = {'batch_size': B, 'points': torch.from_numpy(synthetic.to_batch_points(cl)).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev),
             'points_per_sample_checked': True}
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3, fused=True)
    for _ in range(2):     # first calls build caches (packed weights, grids)
        opt.zero_grad(set_to_none=True)
        ret, tb, disp = model(dict(batch))
        ret['loss'].backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0, foreach=True)
        opt.step()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ret, tb, disp = model(dict(batch))
        ret['loss'].backward()
        # the reference clips every step (tools/train_utils/train_utils.py:58-62): the foreach form keeps the norm on the device
        total = torch.nn.utils.clip_grad_norm_(params, 10.0, foreach=True)
        opt.step()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert torch.isfinite(ret['loss']) and torch.isfinite(total)


@pytest.mark.gpu
def test_detector_eval_returns_nms_filtered_predictions(dev):
    torch.manual_seed(2)
    model = build_pdm_ssd(SMALL).to(dev).eval()
    with torch.no_grad():
        model.point_head.cls_layers[-1].bias.fill_(0.5)      # scores above SCORE_THRESH so NMS has work to do
    B, N = 2, 2048
    pts = torch.from_numpy(synthetic.to_batch_points(synthetic.lidar_like_clouds(B, N, 9))).to(dev)
    with torch.no_grad():
        pred_dicts, recall = model({'batch_size': B, 'points': pts})
    assert len(pred_dicts) == B
    for d in pred_dicts:
        n = d['pred_boxes'].shape[0]
        assert 0 < n <= 500 and d['pred_boxes'].shape[1] == 7 and d['pred_scores'].shape == (n,) and d['pred_labels'].shape == (n,)
        assert bool((d['pred_scores'][:-1] >= d['pred_scores'][1:]).all()) and bool((d['pred_scores'] >= 0.1).all())
        assert int(d['pred_labels'].min()) >= 1 and int(d['pred_labels'].max()) <= 3
        from pdm_ssd_amd.iou3d_nms import iou3d_nms_utils as iu
        iou = iu.boxes_iou_bev(d['pred_boxes'].contiguous(), d['pred_boxes'].contiguous())
        iou.fill_diagonal_(0)
        assert float(iou.max()) <= 0.1 + 1e-4                # survivors do not overlap beyond NMS_THRESH


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cap", [(2, 188, 188, 2), (1, 93, 90, 2), (3, 64, 48, 1), (1, 200, 176, 12), (2, 47, 97, 1)])
def test_heatmap_head_one_kernel_equals_two_kernels_on_odd_maps(dev, B, H, W, cap):
    """pdm_bev_head_fused (depthwise prologue + per-cell stack in one kernel) against the depthwise kernel followed by the row MLP:
    bit-equal logits on maps whose sides are not multiples of the 4 x 16 patch, with one, a few and many tiles per workgroup
    (grid cap), sparse maps with a border."""
    from pdm_ssd_amd import _native
    torch.manual_seed(B * 1000 + H + W)
    head = build_pdm_ssd().dense_head.to(dev).eval()
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    x = torch.randn(B, H, W, 128, device=dev) * (torch.rand(B, H, W, 1, device=dev) < 0.4)
    sf = x.permute(0, 3, 1, 2)
    l = _native.lib()
    oldcap = l.pdm_tune_rows_chain_dw_wg_per_cu(cap)
    try:
        with torch.no_grad():
            head({'spatial_features': sf})
            one = head.forward_ret_dict['hm_logits'].clone()
            head.use_one_kernel = False
            head({'spatial_features': sf})
            two = head.forward_ret_dict['hm_logits'].clone()
            head.use_one_kernel = True
    finally:
        l.pdm_tune_rows_chain_dw_wg_per_cu(oldcap)
    assert tuple(one.shape) == (B, 3, H, W) and torch.isfinite(one).all()
    assert torch.equal(one, two)


@pytest.mark.gpu
@pytest.mark.parametrize("B,M,seed", [(4, 12, 0), (2, 40, 1), (1, 1, 2), (2, 0, 3)])
def test_heatmap_targets_kernel_equals_torch_formulation(dev, B, M, seed):
    """pdm_heatmap_targets against assign_targets' torch formulation (gaussian_radius + one batched scatter-max): the same
    map bit for bit, with padding rows, boxes on the map's border, coinciding centres, tiny boxes (MIN_RADIUS) and large ones
    (window clipped at MAX_RADIUS)."""
    head = build_pdm_ssd().dense_head.to(dev).train()
    rng = np.random.default_rng(seed)
    gt = scene_boxes(B, max(M, 1), seed)[:, :M].copy() if M else np.zeros((B, 0, 8), np.float32)
    if M >= 6:
        gt[:, 1] = 0.0                                              # a padding row in the middle
        gt[:, 2, :2] = [0.05, -39.9]                                # on the map's corner
        gt[:, 3, :3] = gt[:, 4, :3]; gt[:, 3, 7] = gt[:, 4, 7]      # two boxes of one class on one cell
        gt[:, 5, 3:5] = [0.3, 0.2]                                  # smaller than a cell: MIN_RADIUS
        gt[0, 0, 3:5] = [30.0, 12.0]                                # a radius beyond MAX_RADIUS
    g = torch.from_numpy(gt.astype(np.float32)).to(dev)
    H, W = 188, 188
    a = head.assign_targets(g, (H, W))
    head.use_fused_loss = False
    b = head.assign_targets(g, (H, W))
    head.use_fused_loss = True
    assert a.shape == b.shape == (B, 3, H, W) and a.dtype == torch.float32
    assert torch.equal(a, b)
    if M >= 6:
        assert int((a == 1).sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("bf16,layout", [(False, "nchw"), (False, "nhwc"), (True, "nhwc")])
def test_heatmap_focal_loss_kernel_equals_torch_formulation(dev, bf16, layout):
    """pdm_heatmap_focal_loss (clamped sigmoid + penalty-reduced focal loss + gradient) against the torch formulation under
    autograd: loss 1e-5, gradient 1e-5 of its scale (fp32) / one bf16 rounding (bf16 logits); saturated logits (the clamp's
    zero-gradient zone), a map without peaks, strided logits."""
    from pdm_ssd_amd import heatmap_loss
    from pdm_ssd_amd.utils import loss_utils
    torch.manual_seed(7)
    B, C, H, W = 3, 3, 61, 47
    head = build_pdm_ssd().dense_head
    gt = torch.from_numpy(scene_boxes(B, 9, 5)).to(dev)
    hm = heatmap_loss.heatmap_targets(gt, C, H, W, 0.0, -40.0, 0.4, 0.4, 1, 0.1, 2, 8)
    for case in ("peaks", "no peaks"):
        if case == "no peaks":
            hm = torch.where(hm == 1, torch.full_like(hm, 0.5), hm)
        x = torch.randn(B, C, H, W, device=dev) * 4
        x[0, 0, :4, :4] = 30.0; x[0, 1, :4, :4] = -30.0            # clamp(sigmoid) saturates: no gradient there
        if layout == "nhwc":
            x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        if bf16:
            x = x.bfloat16()
        x1 = x.clone().requires_grad_(True)
        x2 = x.clone().requires_grad_(True)
        la = heatmap_loss.heatmap_focal_loss(x1, hm, 1.5)
        lb = loss_utils.neg_loss_cornernet(head.sigmoid(x2.float()), hm) * 1.5
        (la * 2.0).backward(); (lb * 2.0).backward()
        assert la.dtype == torch.float32 and la.dim() == 0
        assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb))), (case, float(la), float(lb))
        ga, gb = x1.grad.float(), x2.grad.float()
        assert x1.grad.dtype == x.dtype and x1.grad.shape == x.shape
        scale = float(gb.abs().max())
        tol = (1.0 / 128 if bf16 else 1e-5) * scale
        assert float((ga - gb).abs().max()) <= tol, (case, float((ga - gb).abs().max()), scale)
        assert float(ga[0, 0, :4, :4].abs().max()) == 0.0 and float(ga[0, 1, :4, :4].abs().max()) == 0.0


@pytest.mark.gpu
def test_heatmap_head_fused_inference_equals_torch_layers(dev):
    """pdm_bev_depthwise3x3 + the per-cell MFMA row kernels against the torch convolutions of the same module, on a
    channels-last grid like the neck's (with empty cells and a border)."""
    torch.manual_seed(3)
    head = build_pdm_ssd().dense_head.to(dev).eval()
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    x = torch.randn(3, 200, 176, 128, device=dev) * (torch.rand(3, 200, 176, 1, device=dev) < 0.3)
    sf = x.permute(0, 3, 1, 2)                                   # (B, C, H, W) view of channels-last storage
    with torch.no_grad():
        a = head({'spatial_features': sf})['bev_heatmap'].clone()
        la = head.forward_ret_dict['hm_logits'].clone()
        assert head._pdm_fused_cache['pw'][1] is not None
        head.use_fused = False
        b = head({'spatial_features': sf})['bev_heatmap']
        lb = head.forward_ret_dict['hm_logits']
        head.use_fused, head.use_one_kernel = True, False            # depthwise kernel, then the per-cell row MLP
        lc = head({'spatial_features': sf}) and head.forward_ret_dict['hm_logits']
    assert tuple(a.shape) == (3, 3, 200, 176)
    assert torch.equal(la, lc)                                       # one launch == two launches, bit for bit
    torch.testing.assert_close(la, lb, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_depthwise3x3_hip_forward_backward_match_torch_conv(dev):
    """pdm_bev_depthwise3x3 (+ its mirrored-tap data gradient and pdm_bev_depthwise3x3_wgrad) against
    torch.nn.functional.conv2d(groups=C) in fp32: forward 1e-5, gradients 1e-4 (float atomics in the weight sum)."""
    import torch.nn.functional as F
    from pdm_ssd_amd.dense_heads.pdm_heatmap_head import _Depthwise3x3CL
    torch.manual_seed(4)
    for (B, C, H, W) in ((2, 128, 50, 44), (1, 8, 5, 3), (3, 64, 17, 200)):
        x = (torch.randn(B, H, W, C, device=dev) * (torch.rand(B, H, W, 1, device=dev) < 0.5)).permute(0, 3, 1, 2)
        w = torch.randn(C, 1, 3, 3, device=dev)
        x1, w1 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
        x2, w2 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
        want = F.conv2d(x1, w1, padding=1, groups=C)
        got = _Depthwise3x3CL.apply(x2, w2)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
        go = torch.randn_like(want)
        want.backward(go)
        got.backward(go)
        torch.testing.assert_close(x2.grad, x1.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(w2.grad, w1.grad, rtol=1e-4, atol=1e-4 * float(w1.grad.abs().max()))


@pytest.mark.gpu
def test_depthwise3x3_with_a_bf16_output_is_the_fp32_operator_rounded_once(dev):
    """_Depthwise3x3CL(out_bf16=True) (pdm_bev_depthwise3x3_t / _wgrad_t: the training form under bf16 autocast): the output is
    the fp32 operator's output rounded to bf16, BIT for bit; fed a bf16 gradient, the input gradient equals the fp32 operator's
    on that gradient bit for bit (same kernel arithmetic, bf16 values are exact in fp32) and the weight gradient agrees to the
    order of the float atomics.  Odd sizes (one-cell kernel) included."""
    from pdm_ssd_amd.dense_heads.pdm_heatmap_head import _Depthwise3x3CL
    torch.manual_seed(6)
    for (B, C, H, W) in ((2, 128, 50, 44), (1, 8, 5, 3), (3, 64, 17, 200)):
        x = (torch.randn(B, H, W, C, device=dev) * (torch.rand(B, H, W, 1, device=dev) < 0.5)).permute(0, 3, 1, 2)
        w = torch.randn(C, 1, 3, 3, device=dev)
        x1, w1 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
        x2, w2 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
        want = _Depthwise3x3CL.apply(x1, w1)
        got = _Depthwise3x3CL.apply(x2, w2, True)
        assert got.dtype == torch.bfloat16 and torch.equal(got, want.to(torch.bfloat16))
        go = torch.randn_like(want).to(torch.bfloat16)
        want.backward(go.float())
        got.backward(go)
        assert torch.equal(x2.grad, x1.grad)
        torch.testing.assert_close(w2.grad, w1.grad, rtol=1e-4, atol=1e-4 * float(w1.grad.abs().max()))
