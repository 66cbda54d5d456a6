"""BASELINE.json configs[4] and configs[3] on the GPU, through the C ABI, against the CPU oracle.

configs[4] — dense scene stress, 65536 points per cloud: every operator of the path at that size (multi-workgroup
FPS, grid ball query, grid three-NN, neighbour-list compaction + packed SA kernels, the whole PointNet2MSG forward,
the PDM neck's ATOMICS path on that backbone's sampled set, and the pipelined step with the level-1 FPS cut into
resumable segments of cooperating workgroups) against oracle/pointnet2_oracle.c / pdm_oracle.c / the CPU graph.
Semantics: group_points_gpu.cu:14-31,53-72, ball_query_gpu.cu:15-51, interpolate_gpu.cu:16-59,127-149 of
/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/.

configs[3] — training step: backbone + neck forward/backward under bf16 autocast against the fp32 CPU autograd
graph built on the oracle operators (oracle/cpu_autograd.py).  Tolerances are stated at the asserts.
"""
import copy

import numpy as np
import pytest
import torch

from pdm_ssd_amd import _native, pdm_ops, synthetic
from pdm_ssd_amd.pdm_neck import PDMNeck
from pdm_ssd_amd.pointnet2_backbone import POINTRCNN_MSG_CFG, PointNet2MSG
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu

pytestmark = pytest.mark.gpu

N_BIG, B_BIG = 65536, 2
NECK_CFG = {'SOURCE_LAYER': 2, 'FEATURE_DIM': 128, 'DILATION': [7, 7, 1], 'SH_DEGREE': 2, 'BEV_STRIDE': 8,
            'HEIGHT_BINS': 1, 'INPUT_CHANNELS': 256, 'NORMALIZE': True}


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def randomize_bn(module, seed):
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


def make_neck(cfg=NECK_CFG, seed=0):
    torch.manual_seed(seed)
    neck = PDMNeck(cfg, grid_size=[1408, 1600, 40], voxel_size=[0.05, 0.05, 0.1],
                   point_cloud_range=list(synthetic.KITTI_RANGE))
    with torch.no_grad():
        neck.coef.weight.normal_(0.0, 0.02)
    return neck


@pytest.fixture(scope="module")
def big(oracle):
    """Two lidar-like clouds of 65536 points, the oracle's level-1 FPS (65536 -> 4096) and sampled coordinates."""
    cl = synthetic.lidar_like_clouds(B_BIG, N_BIG, 4242)
    xyz = np.ascontiguousarray(cl[:, :, :3])
    idx = oracle.furthest_point_sample(xyz, 4096)
    new_xyz = np.ascontiguousarray(np.take_along_axis(xyz, idx[:, :, None].astype(np.int64), 1))
    return {'clouds': cl, 'xyz': xyz, 'fps_idx': idx, 'new_xyz': new_xyz}


# ------------------------------------------------------------------ configs[4]: operators at 65536 points

def test_fps_65536_to_4096_index_exact(big, dev):
    got = pu.furthest_point_sample(T(big['xyz'], dev), 4096)
    _native.fps_check()   # no cooperating workgroup gave up waiting for a peer
    np.testing.assert_array_equal(got.cpu().numpy(), big['fps_idx'])


@pytest.mark.parametrize("radius,nsample", [(0.1, 16), (0.5, 32)])
def test_ball_query_grid_65536_index_exact(oracle, big, dev, radius, nsample):
    ref = oracle.ball_query(radius, nsample, big['xyz'], big['new_xyz'])
    got = pu.ball_query(radius, nsample, T(big['xyz'], dev), T(big['new_xyz'], dev))
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_three_nn_65536_index_and_distance_exact(oracle, big, dev):
    ref_d, ref_i = oracle.three_nn(big['xyz'], big['new_xyz'])
    d, i = pu.three_nn(T(big['xyz'], dev), T(big['new_xyz'], dev))
    np.testing.assert_array_equal(i.cpu().numpy(), ref_i)
    np.testing.assert_array_equal(d.cpu().numpy(), ref_d)


def test_group_points_and_grad_65536(oracle, big, dev):
    """group_points at the stress size: forward bit-exact, backward (scatter-add, group_points_gpu.cu:14-31) to 1e-4."""
    idx = oracle.ball_query(0.5, 32, big['xyz'], big['new_xyz'])
    rng = np.random.default_rng(3)
    feat = rng.standard_normal((B_BIG, 6, N_BIG)).astype(np.float32)
    f = T(feat, dev).requires_grad_(True)
    out = pu.grouping_operation(f, T(idx, dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.grouping_operation(feat, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(T(go, dev))
    ref = oracle.grouping_operation_grad(go, idx, N_BIG)
    np.testing.assert_allclose(f.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


def test_sa_pack_and_packed_sa_65536(big, dev):
    """Level 1 of PointNet2MSG on 65536-point clouds: compacted neighbour lists + packed SA kernels == dense kernels
    bit for bit, and both within 1e-4 of the CPU graph (oracle operators + torch-CPU MLP)."""
    from oracle import cpu_backbone
    torch.manual_seed(3)
    sa = PointNet2MSG(POINTRCNN_MSG_CFG, input_channels=4).SA_modules[0].eval()
    randomize_bn(sa, 5)
    feat = np.ascontiguousarray(big['clouds'][:, :, 3:].transpose(0, 2, 1))
    ref_xyz, ref_feat = cpu_backbone.sa_forward(sa, big['xyz'], feat)
    np.testing.assert_array_equal(ref_xyz, big['new_xyz'])
    g = copy.deepcopy(sa).to(dev)
    x, f = T(big['xyz'], dev), T(feat, dev)
    with torch.no_grad():
        nx, packed = g(x, f)
        g.use_pack = False
        _, dense = g(x, f)
    np.testing.assert_array_equal(nx.cpu().numpy(), ref_xyz)
    assert torch.equal(packed, dense)
    np.testing.assert_allclose(packed.cpu().numpy(), ref_feat, rtol=1e-4, atol=1e-4)


@pytest.fixture(scope="module")
def big_backbone(big, dev):
    """PointNet2MSG (pointrcnn config) forward on the two 65536-point clouds: GPU result and CPU-graph reference."""
    from oracle import cpu_backbone
    torch.manual_seed(7)
    bb = PointNet2MSG(POINTRCNN_MSG_CFG, input_channels=4).eval()
    randomize_bn(bb, 8)
    ref = cpu_backbone.backbone_forward(bb, big['clouds'])
    bb_g = copy.deepcopy(bb).to(dev)
    pts = T(synthetic.to_batch_points(big['clouds']), dev)
    with torch.no_grad():
        bd = bb_g({'batch_size': B_BIG, 'points': pts})
    _native.fps_check()
    return {'cpu': bb, 'gpu': bb_g, 'ref': ref, 'bd': bd, 'points': pts}


def test_backbone_65536_matches_cpu_graph(big_backbone):
    bd, ref = big_backbone['bd'], big_backbone['ref']
    for k in range(1, 5):
        np.testing.assert_array_equal(bd['sa_xyz'][k].cpu().numpy(), ref['sa_xyz'][k])
        np.testing.assert_allclose(bd['sa_features'][k].cpu().numpy(), ref['sa_features'][k], rtol=1e-4, atol=1e-4)
    assert tuple(bd['point_features'].shape) == (B_BIG * N_BIG, 128)
    np.testing.assert_allclose(bd['point_features'].cpu().numpy(), ref['point_features'], rtol=1e-4, atol=1e-4)


def test_pdm_atomics_path_on_stress_backbone(oracle, big_backbone, dev):
    """The PDM neck's scatter-add (atomics-on-HBM, pdm_scatter_bev) on the sampled set of the 65536-point backbone,
    bench-sized grid (128 channels, 7x7 dilation, degree 2, 176 x 200 cells): against oracle/pdm_oracle.c on the same
    inputs (fp32 atomic order is free: 1e-4 of the grid's scale) and against the gather form the inference path uses."""
    bd = big_backbone['bd']
    neck = make_neck().to(dev).eval()
    xyz = bd['sa_xyz'][2].contiguous()
    src = bd['sa_features'][2].contiguous()
    with torch.no_grad():
        feat = neck.proj(src).transpose(1, 2).contiguous()
        co = neck.coef(src).transpose(1, 2)
        sh = co[..., :neck.nsh].contiguous()
        sigma = torch.nn.functional.softplus(co[..., neck.nsh]) + neck.sigma_min
        inv2s2 = (0.5 / (sigma * sigma)).contiguous()
        grid, wsum = pdm_ops.pdm_scatter(xyz, feat, sh, inv2s2, neck.grid, neck.dilation, neck.degree, 1)
        ggrid, gwsum = pdm_ops.pdm_gather(xyz, feat, sh, inv2s2, neck.grid, neck.dilation, neck.degree, normalize=False)
    g = neck.grid
    ref_grid, ref_wsum = oracle.pdm_scatter(xyz.cpu().numpy(), feat.cpu().numpy(), sh.cpu().numpy(), inv2s2.cpu().numpy(),
                                            g.origin, g.cell, g.inv_cell, (g.W, g.H, g.D), neck.dilation, neck.degree, layout=1)
    assert tuple(grid.shape) == (B_BIG, 200, 176, 128)
    scale = float(np.abs(ref_grid).max())
    assert scale > 0 and float((ref_wsum != 0).mean()) > 0.01      # the dilated points really cover part of the grid
    np.testing.assert_allclose(grid.cpu().numpy(), ref_grid, rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(wsum.cpu().numpy(), ref_wsum, rtol=1e-4, atol=1e-4 * float(np.abs(ref_wsum).max()))
    np.testing.assert_allclose(ggrid.cpu().numpy(), ref_grid, rtol=1e-4, atol=1e-4 * scale)
    # the module's two paths (training: atomics + normalise; inference: gather with the normalisation fused)
    with torch.no_grad():
        a = neck(dict(bd))['spatial_features'].clone()
        neck.use_gather = False
        b = neck(dict(bd))['spatial_features']
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(a.abs().max()))


def test_pipelined_step_65536_depth4_equals_serial(big_backbone, dev):
    """65536 points x pipeline depth 4: the level-1 FPS runs as three resumable segments of four cooperating
    workgroups per cloud, three batches side by side in one launch; every batch still comes out bit for bit as from
    the serial path."""
    from pdm_ssd_amd.pipeline import PipelinedHotPath
    backbone = big_backbone['gpu']
    neck = make_neck().to(dev).eval()
    depth, S = 4, 3
    batches = [big_backbone['points']] + [
        T(synthetic.to_batch_points((synthetic.lidar_like_clouds if i % 2 else synthetic.uniform_clouds)(B_BIG, N_BIG, 500 + i)), dev)
        for i in range(1, 2 + S + 1)]
    with torch.no_grad():
        ref = []
        for p in batches[:2]:
            bd = neck(backbone({'batch_size': B_BIG, 'points': p}))
            ref.append((bd['point_features'].clone(), bd['spatial_features'].clone()))
        pipe = PipelinedHotPath(backbone, neck, depth=depth)
        pipe.prime_segmented(batches[:S + 1], B_BIG)
        for i in range(2):
            bd = pipe.step(batches[i], None, B_BIG, points_ahead=batches[i + 1:i + 2 + S])
            torch.cuda.synchronize()
            pipe.check_sampling()
            assert torch.equal(bd['point_features'], ref[i][0]), f"batch {i}"
            assert torch.equal(bd['spatial_features'], ref[i][1]), f"batch {i}"


def test_fps_status_word_raises(dev):
    """A cooperating-workgroup FPS that gives up waiting for a peer raises its status word; the host check turns that
    into an exception (forced here by setting the word by hand), and a normal call leaves it clear."""
    lib = _native.lib()
    assert lib.pdm_fps_max_coresident_workgroups() >= 8
    b, n = 2, 40000
    nbytes = lib.pdm_furthest_point_sampling_ws_bytes(b, n)
    xyz = T(synthetic.uniform_clouds(b, n, 9)[:, :, :3], dev)
    temp = torch.full((b, n), 1e10, device=dev)
    idx = torch.empty((b, 64), dtype=torch.int32, device=dev)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _native.call("pdm_furthest_point_sampling_ws", torch.cuda.current_stream().cuda_stream, b, n, 64, xyz.data_ptr(),
                 temp.data_ptr(), idx.data_ptr(), ws.data_ptr(), nbytes)
    _native.fps_check_workspace(ws, b, n)          # clear after a normal call
    ws[nbytes - 64] = 1
    with pytest.raises(_native.NativeLibraryError, match="gave up waiting"):
        _native.fps_check_workspace(ws, b, n)


def test_backward_at_exactly_16384_source_points(dev):
    """N == 16384 with features that require grad: the CSR build kernels of the channels-last grouping backward and of
    the three_interpolate backward take 64 KB + of LDS there (dynamic n * 4 bytes + a static block)."""
    torch.manual_seed(0)
    xyz = T(synthetic.lidar_like_clouds(1, 16384, 77)[:, :, :3], dev)
    new_xyz = xyz[:, :512].contiguous()
    f1 = torch.randn(1, 8, 16384, device=dev, requires_grad=True)
    f2 = f1.detach().clone().requires_grad_(True)
    ref_mod, cl_mod = pu.QueryAndGroup(0.8, 16), pu.QueryAndGroup(0.8, 16)
    cl_mod.channels_last = True
    want, got = ref_mod(xyz, new_xyz, f1), cl_mod(xyz, new_xyz, f2)
    go = torch.randn_like(want)
    want.backward(go)
    got.backward(go.contiguous(memory_format=torch.channels_last))
    torch.testing.assert_close(f2.grad, f1.grad, rtol=1e-4, atol=1e-4)
    known = torch.randn(1, 5, 16384, device=dev, requires_grad=True)
    idx = torch.randint(0, 16384, (1, 3000, 3), device=dev, dtype=torch.int32)
    w = torch.rand(1, 3000, 3, device=dev)
    out = pu.three_interpolate(known, idx, w)
    out.backward(torch.ones_like(out))
    ref = torch.zeros(5, 16384, device=dev)
    for k in range(3):
        ref.index_add_(1, idx[0, :, k].long(), w[0, :, k].expand(5, -1).contiguous())
    torch.testing.assert_close(known.grad[0], ref, rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ configs[3]: the training step

TRAIN_CFG = {
    'SA_CONFIG': {'NPOINTS': [512, 128, 32], 'RADIUS': [[0.5, 1.0], [1.0, 2.0], [2.0, 4.0]],
                  'NSAMPLE': [[16, 32], [16, 32], [16, 32]],
                  'MLPS': [[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]], [[128, 196, 256], [128, 196, 256]]]},
    'FP_MLPS': [[128, 128], [256, 256], [512, 512]],
}
TRAIN_NECK = {'SOURCE_LAYER': 2, 'FEATURE_DIM': 32, 'DILATION': [5, 5, 1], 'SH_DEGREE': 2, 'BEV_STRIDE': 16,
              'HEIGHT_BINS': 1, 'INPUT_CHANNELS': 256, 'NORMALIZE': True}


def _train_loss(pf, sf):
    return pf.float().square().mean() + sf.float().square().mean()


@pytest.fixture(scope="module")
def train_reference():
    """fp32 CPU autograd graph on the oracle operators: loss and per-parameter gradients of one training step."""
    from oracle import cpu_autograd
    torch.manual_seed(11)
    bb = PointNet2MSG(TRAIN_CFG, input_channels=4).train()
    neck = make_neck(TRAIN_NECK, seed=12).train()
    clouds = synthetic.lidar_like_clouds(2, 2048, 31)
    bb_c, neck_c = copy.deepcopy(bb), copy.deepcopy(neck)
    out = cpu_autograd.train_forward(bb_c, neck_c, clouds)
    loss = _train_loss(out['point_features'], out['spatial_features'])
    loss.backward()
    grads = {f"backbone.{k}": p.grad.clone() for k, p in bb_c.named_parameters()}
    grads.update({f"neck.{k}": p.grad.clone() for k, p in neck_c.named_parameters()})
    return {'bb': bb, 'neck': neck, 'clouds': clouds, 'loss': float(loss), 'grads': grads,
            'pf': out['point_features'].detach(), 'sf': out['spatial_features'].detach()}


def _gpu_train_step(ref, dev, autocast):
    bb, neck = copy.deepcopy(ref['bb']).to(dev).train(), copy.deepcopy(ref['neck']).to(dev).train()
    pts = T(synthetic.to_batch_points(ref['clouds']), dev)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        bd = neck(bb({'batch_size': ref['clouds'].shape[0], 'points': pts}))
        loss = _train_loss(bd['point_features'], bd['spatial_features'])
    loss.backward()
    grads = {f"backbone.{k}": p.grad.float().cpu() for k, p in bb.named_parameters()}
    grads.update({f"neck.{k}": p.grad.float().cpu() for k, p in neck.named_parameters()})
    return float(loss), grads, bd


def _grad_errors(got, want):
    """relative L2 error per parameter, measured against the larger of the parameter's own gradient norm and 1e-3 of
    the largest gradient norm of the model (a gradient that is ~0 in exact arithmetic has no relative error)."""
    floor = 1e-3 * max(float(w.norm()) for w in want.values())
    return {k: float((got[k] - w).norm()) / max(float(w.norm()), floor) for k, w in want.items()}


def test_fp32_train_step_matches_cpu_autograd(train_reference, dev):
    """Same step in fp32 on the GPU (HIP operators + their backward kernels, MIOpen/rocBLAS layers): loss to 1e-4,
    every parameter gradient to 2e-3 relative L2 (fp32 summation order only)."""
    loss, grads, bd = _gpu_train_step(train_reference, dev, autocast=False)
    assert abs(loss - train_reference['loss']) <= 1e-4 * abs(train_reference['loss'])
    np.testing.assert_allclose(bd['point_features'].detach().cpu().numpy(), train_reference['pf'].numpy(), rtol=1e-3, atol=1e-3)
    err = _grad_errors(grads, train_reference['grads'])
    assert set(grads) == set(train_reference['grads'])
    worst = max(err, key=err.get)
    assert err[worst] <= 2e-3, (worst, err[worst])


def test_bf16_train_step_matches_fp32_cpu_autograd(train_reference, dev):
    """BASELINE configs[3]: the bf16-autocast training step (bf16 shared MLPs and grouped tensors, fp32 coordinates /
    indices / operators) against the fp32 CPU autograd graph.  Stated bf16 tolerance: loss within 2 %; the whole
    gradient vector within 50 % relative L2 and every parameter gradient with cosine >= 0.8 against fp32; the neck's
    gradients (two layers from the loss) within 30 %.  Why so wide: the same step in fp32 matches the CPU graph to 1e-4
    (test above), and torch's OWN bf16 autocast of these layers over fp32 operators (PDM_CHANNELS_LAST=0) sits at
    0.26 relative L2 against 0.31 for this path (tools/diag/bf16_grad_table.py) — the error is bf16 BatchNorm / conv
    arithmetic under a max-pool, not the operators, whose bf16 forms are held bit-exact / 2e-2 per operator in
    tests/test_modules_gpu.py::test_query_and_group_channels_last.  The figure moves with the shape and with which
    BatchNorm runs: neck.coef.weight 0.04 / 0.05 / 0.39 through torch's BatchNorm and 0.21 / 0.05 / 0.67 through the
    fused one (pdm_ssd_amd/fused_bn.py) at (B, N) = (2, 2048) / (3, 3000) / (2, 8192), while per operator the fused
    BatchNorm is the more accurate of the two against fp64 (tools/diag/bn_error_stats.py: MIOpen truncates its bf16
    outputs, 2x the mean error and a -1e-3 bias)."""
    loss, grads, bd = _gpu_train_step(train_reference, dev, autocast=True)
    want = train_reference['grads']
    assert abs(loss - train_reference['loss']) <= 2e-2 * abs(train_reference['loss']), (loss, train_reference['loss'])
    assert bd['point_features'].dtype == torch.bfloat16          # the MLPs really ran under autocast
    for k, g in grads.items():
        assert torch.isfinite(g).all(), k
    err = _grad_errors(grads, want)
    floor = 1e-3 * max(float(w.norm()) for w in want.values())
    for k, w in want.items():
        if float(w.norm()) > floor:
            cos = float((grads[k] * w).sum() / (grads[k].norm() * w.norm()))
            assert cos >= 0.8, (k, cos)
        if k.startswith("neck."):
            assert err[k] <= 0.3, (k, err[k])
    flat_g = torch.cat([grads[k].reshape(-1) for k in want])
    flat_w = torch.cat([want[k].reshape(-1) for k in want])
    assert float((flat_g - flat_w).norm() / flat_w.norm()) <= 0.5


# ------------------------------------------------------------------ configs[3]: the FULL detector's training step
# backbone + PDM neck + hybrid head with the detector's real losses (point focal + smooth-L1 over points_in_boxes
# targets, heat-map focal: dense_heads/point_head_template.py:82-89,127-183, utils/loss_utils.py:335-345,
# detectors/point_rcnn.py:13-30 of the reference), against oracle/cpu_detector.py.

@pytest.fixture(scope="module")
def detector_reference():
    from detector_case import build_case
    from oracle import cpu_detector
    model, clouds, gt = build_case()
    return {'model': model, 'clouds': clouds, 'gt': gt,
            'fp32': cpu_detector.detector_train_step(model, clouds, gt, bf16=False),
            'bf16': cpu_detector.detector_train_step(model, clouds, gt, bf16=True)}


def _gpu_detector_step(ref, dev, autocast):
    model = copy.deepcopy(ref['model']).to(dev).train()
    batch = {'batch_size': ref['clouds'].shape[0], 'points': T(synthetic.to_batch_points(ref['clouds']), dev),
             'gt_boxes': T(ref['gt'], dev)}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        for module in model.module_list:                      # the detector's forward loop (detectors/pdm_ssd.py), batch dict kept
            batch = module(batch)
        loss, tb, _ = model.get_training_loss()
    loss.backward()
    grads = {k: p.grad.float().cpu() for k, p in model.named_parameters()}
    fr = dict(model.point_head.forward_ret_dict)
    fr.update(sa_features=batch['sa_features'], point_features=batch['point_features'], spatial_features=batch['spatial_features'],
              hm_logits=model.dense_head.forward_ret_dict['hm_logits'])
    return float(loss), {k: float(v) for k, v in tb.items()}, grads, fr, model


def _dump_grad_table(name, err, extra=None):
    """per-parameter error table under gpurun_out/ (travels back from the GPU box): what the stated bounds rest on"""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w") as f:
        json.dump({'errors': dict(sorted(err.items(), key=lambda kv: -kv[1])), 'extra': extra}, f, indent=1)


def test_fp32_detector_train_step_matches_cpu_graph(detector_reference, dev):
    """fp32 GPU step of the whole detector (HIP operators and their backward kernels, fused BatchNorm kernels,
    points_in_boxes targets, MIOpen / rocBLAS layers) against the CPU graph on the oracle operators: loss and every
    loss term to 1e-4, labels identical, every parameter gradient to 4e-3 relative L2, the median below 1.5e-3.
    (fp32 summation order on BOTH sides: two CPU statements of this graph — torch layers vs the hand-written graph of
    oracle/cpu_detector.py, unrounded — already differ by up to 7e-4, median 3e-4, tests/test_cpu_autograd.py; measured
    GPU vs CPU: median 7e-4, worst 2.4e-3 on a BatchNorm bias of the point head, a sum of 4096 signed terms.)"""
    from detector_case import grad_errors
    want = detector_reference['fp32']
    loss, tb, grads, fr, _ = _gpu_detector_step(detector_reference, dev, autocast=False)
    assert abs(loss - want['loss']) <= 1e-4 * abs(want['loss']), (loss, want['loss'])
    for k, v in want['tb'].items():
        assert abs(tb[k] - v) <= 1e-4 * max(abs(v), 1.0), (k, tb[k], v)
    assert torch.equal(fr['point_cls_labels'].cpu(), want['point_cls_labels'])
    np.testing.assert_allclose(fr['point_cls_preds'].detach().float().cpu().numpy(), want['point_cls_preds'].numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(fr['hm_logits'].detach().float().cpu().numpy(), want['hm_logits'].numpy(), rtol=1e-3, atol=1e-3)
    assert set(grads) == set(want['grads'])
    err = grad_errors(grads, want['grads'])
    _dump_grad_table("detector_step_fp32_grad_errors.json", err)
    worst = max(err, key=err.get)
    assert err[worst] <= 4e-3, (worst, err[worst])
    assert float(np.median(list(err.values()))) <= 1.5e-3


# Stated bf16 tolerances of the detector step against the bf16-EMULATING CPU graph (measured values:
# profiles/r03_detector_step_bf16_grad_errors.json): see the asserts.
BF16_LOSS_TOL = 1e-3
BF16_GRAD_TOL = 0.1
BF16_GRAD_MEDIAN_TOL = 0.06


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def test_bf16_detector_train_step_matches_bf16_emulating_cpu_graph(detector_reference, dev):
    """BASELINE configs[3]: the bf16-autocast training step of the whole detector (every 1x1 convolution / Linear on this
    library's bf16 MFMA kernels, fused BatchNorm, HIP operators) against a CPU graph that rounds to bf16 at the same
    points, forward and backward (oracle/cpu_detector.py): bf16 against bf16.
    Forward: the SA stack, the neck's grid and the heat-map logits equal the emulation to 1e-3 relative L2 (measured
    0 - 1e-4: the only difference is the fp32 summation order inside a contraction, which moves a bf16 rounding where the
    sum sits on a tie); point features / logits behind the three FP levels to 3e-2 (measured 8e-3 / 1.4e-2).
    Loss to 1e-3 (measured 6e-5).  Every parameter gradient to 0.1 relative L2, the median to 0.06 (measured worst 0.064,
    median 0.036): in the backward every flipped rounding of an input gradient feeds the layers below it, so the
    distance grows towards the first layers.  Against the fp32 graph the same gradients sit at 0.27 - 0.87 (recorded in
    the table), which is why round 2's comparison needed a 0.5 whole-vector bound."""
    from detector_case import grad_errors
    want, want32 = detector_reference['bf16'], detector_reference['fp32']
    loss, tb, grads, fr, model = _gpu_detector_step(detector_reference, dev, autocast=True)
    assert fr['point_cls_preds'].dtype == torch.bfloat16          # the layers really ran under autocast
    assert torch.equal(fr['point_cls_labels'].cpu(), want['point_cls_labels'])
    fwd = {f'sa_features[{i}]': _rel(fr['sa_features'][i], want['sa_features'][i]) for i in (1, 2, 3)}
    fwd.update({k: _rel(fr[k], want[k]) for k in ('spatial_features', 'hm_logits', 'point_features', 'point_cls_preds', 'point_box_preds')})
    for k, g in grads.items():
        assert torch.isfinite(g).all(), k
    err = grad_errors(grads, want['grads'])
    err32 = grad_errors(grads, want32['grads'])
    emu_vs_32 = grad_errors(want['grads'], want32['grads'])
    _dump_grad_table("detector_step_bf16_grad_errors.json", err,
                     {'loss_gpu': loss, 'loss_emulated': want['loss'], 'loss_fp32': want32['loss'], 'tb_gpu': tb, 'tb_emulated': want['tb'],
                      'forward_rel_l2_gpu_vs_emulation': fwd,
                      'gpu_vs_fp32_graph': err32, 'emulation_vs_fp32_graph': emu_vs_32,
                      'median_gpu_vs_emulation': float(np.median(list(err.values()))),
                      'median_gpu_vs_fp32': float(np.median(list(err32.values())))})
    for k in ('sa_features[1]', 'sa_features[2]', 'sa_features[3]', 'spatial_features', 'hm_logits'):
        assert fwd[k] <= 1e-3, (k, fwd[k])
    for k in ('point_features', 'point_cls_preds', 'point_box_preds'):
        assert fwd[k] <= 3e-2, (k, fwd[k])
    assert abs(loss - want['loss']) <= BF16_LOSS_TOL * abs(want['loss']), (loss, want['loss'], want32['loss'])
    worst = max(err, key=err.get)
    assert err[worst] <= BF16_GRAD_TOL, (worst, err[worst], float(np.median(list(err.values()))))
    assert float(np.median(list(err.values()))) <= BF16_GRAD_MEDIAN_TOL
    # and the comparison is the right one: the fp32 graph is several times further away
    assert float(np.median(list(err32.values()))) >= 3 * float(np.median(list(err.values())))
