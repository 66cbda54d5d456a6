"""Module-level parity: SA / FP modules, PointNet2MSG and the PDM neck on the GPU against the CPU
statement of the same graph (oracle operators + torch-CPU MLPs, oracle/cpu_backbone.py).
Indices inside are bit-exact (tested per operator); features are fp32 through different GEMM
implementations (MIOpen/hipBLASLt vs CPU) -> 1e-4, the tolerance north_star states."""
import copy

import numpy as np
import pytest
import torch

from pdm_ssd_amd import synthetic
from pdm_ssd_amd.pdm_neck import PDMNeck
from pdm_ssd_amd.pointnet2_backbone import POINTRCNN_MSG_CFG, PointNet2MSG
from pdm_ssd_amd.pointnet2_batch import pointnet2_modules as pm

pytestmark = pytest.mark.gpu

SMALL_CFG = {
    'SA_CONFIG': {'NPOINTS': [512, 128, 32, 8],
                  'RADIUS': [[0.5, 1.0], [1.0, 2.0], [2.0, 4.0], [4.0, 8.0]],
                  'NSAMPLE': [[16, 32], [16, 32], [16, 32], [8, 16]],
                  'MLPS': [[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]],
                           [[128, 196, 256], [128, 196, 256]], [[256, 256, 512], [256, 384, 512]]]},
    'FP_MLPS': [[128, 128], [256, 256], [512, 512], [512, 512]],
}


def randomize_bn(module, seed):
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)


def test_sa_module_msg(dev):
    from oracle import cpu_backbone
    torch.manual_seed(0)
    sa = pm.PointnetSAModuleMSG(npoint=256, radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[1, 16, 32], [1, 16, 48]]).eval()
    randomize_bn(sa, 1)
    cl = synthetic.lidar_like_clouds(2, 2048, 3)
    xyz = np.ascontiguousarray(cl[:, :, :3]); feat = np.ascontiguousarray(cl[:, :, 3:].transpose(0, 2, 1))
    ref_xyz, ref_feat = cpu_backbone.sa_forward(sa, xyz, feat)
    sa_g = copy.deepcopy(sa).to(dev)
    with torch.no_grad():
        nx, nf = sa_g(torch.from_numpy(xyz).to(dev), torch.from_numpy(feat).to(dev))
    np.testing.assert_array_equal(nx.cpu().numpy(), ref_xyz)
    assert tuple(nf.shape) == (2, 80, 256)
    np.testing.assert_allclose(nf.cpu().numpy(), ref_feat, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("mode", ["eval_grad", "train_nograd", "plain_sequential", "avg_pool"])
def test_sa_module_under_bf16_autocast_outside_the_training_stack(dev, mode):
    """The grouped tensor is born with round8(3 + C) channels only for the consumer that knows about the zero padding
    (TrainSequential's training stack with max-pool).  Eval mode with autograd on, training mode under no_grad, a plain
    nn.Sequential MLP and avg_pool all hand the torch layers exactly 3 + C channels (1 + 3 = 4 and 96 + 3 = 99 here: neither
    a multiple of 8) and must run, with the features of the fp32 graph to bf16 accuracy."""
    torch.manual_seed(0)
    kw = {'pool_method': 'avg_pool'} if mode == "avg_pool" else {}
    sa = pm.PointnetSAModuleMSG(npoint=128, radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[96, 32, 32], [96, 32, 48]], **kw)
    randomize_bn(sa, 3)
    if mode == "plain_sequential":
        sa.mlps = torch.nn.ModuleList(torch.nn.Sequential(*list(m)) for m in sa.mlps)
    sa = sa.to(dev)
    sa.train(mode in ("train_nograd", "plain_sequential", "avg_pool"))
    cl = synthetic.lidar_like_clouds(2, 1024, 3)
    xyz = torch.from_numpy(np.ascontiguousarray(cl[:, :, :3])).to(dev)
    feat = torch.randn(2, 96, 1024, device=dev)
    ref_mod = copy.deepcopy(sa)
    sa.use_fused = False                      # (eval + no_grad would take the fused inference kernels: not the case here)
    ctx = torch.no_grad() if mode == "train_nograd" else torch.enable_grad()
    with ctx, torch.autocast("cuda", dtype=torch.bfloat16):
        nx, nf = sa(xyz, feat)
    assert tuple(nf.shape) == (2, 80, 128) and torch.isfinite(nf.float()).all()
    ref_mod.use_fused = False
    with ctx:
        _, want = ref_mod(xyz, feat)          # the same module in fp32 (same batch statistics in training mode)
    torch.testing.assert_close(nf.float(), want.float(), rtol=0.05, atol=0.05)


def test_fp_module(dev):
    from oracle import cpu_backbone
    torch.manual_seed(1)
    fp = pm.PointnetFPModule(mlp=[40 + 6, 64, 32]).eval()
    randomize_bn(fp, 2)
    rng = np.random.default_rng(0)
    unknown = synthetic.uniform_clouds(2, 1024, 5)[:, :, :3].copy()
    known = np.ascontiguousarray(unknown[:, :200])
    uf = rng.standard_normal((2, 6, 1024)).astype(np.float32)
    kf = rng.standard_normal((2, 40, 200)).astype(np.float32)
    ref = cpu_backbone.fp_forward(fp, unknown, known, uf, kf)
    fp_g = copy.deepcopy(fp).to(dev)
    with torch.no_grad():
        got = fp_g(*(torch.from_numpy(a).to(dev) for a in (unknown, known, uf, kf)))
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


def test_backbone_and_neck_end_to_end(dev):
    from oracle import cpu_backbone
    torch.manual_seed(2)
    bb = PointNet2MSG(SMALL_CFG, input_channels=4).eval()
    randomize_bn(bb, 3)
    neck = PDMNeck({'SOURCE_LAYER': 2, 'FEATURE_DIM': 32, 'DILATION': [5, 5, 1], 'SH_DEGREE': 2, 'BEV_STRIDE': 16,
                    'HEIGHT_BINS': 2, 'INPUT_CHANNELS': 256}, grid_size=[1408, 1600, 40],
                   voxel_size=[0.05, 0.05, 0.1], point_cloud_range=list(synthetic.KITTI_RANGE)).eval()
    with torch.no_grad():
        neck.coef.weight.normal_(0, 0.05)
    randomize_bn(neck, 4)
    cl = synthetic.lidar_like_clouds(2, 2048, 21)
    ref = cpu_backbone.backbone_forward(bb, cl)
    ref_sf = cpu_backbone.neck_forward(neck, ref['sa_xyz'], ref['sa_features'])

    bb_g, neck_g = copy.deepcopy(bb).to(dev), copy.deepcopy(neck).to(dev)
    pts = torch.from_numpy(synthetic.to_batch_points(cl)).to(dev)
    with torch.no_grad():
        bd = neck_g(bb_g({'batch_size': 2, 'points': pts}))
    assert tuple(bd['point_features'].shape) == (2 * 2048, 128)
    assert tuple(bd['point_coords'].shape) == (2 * 2048, 4)
    for k in range(1, 5):
        np.testing.assert_array_equal(bd['sa_xyz'][k].cpu().numpy(), ref['sa_xyz'][k])
    np.testing.assert_allclose(bd['point_features'].cpu().numpy(), ref['point_features'], rtol=1e-4, atol=1e-4)
    sf = bd['spatial_features']
    assert tuple(sf.shape) == (2, 64, 100, 88) and bd['spatial_features_stride'] == 16
    np.testing.assert_allclose(sf.cpu().numpy(), ref_sf, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(ref_sf).max()))


def test_backbone_rejects_ragged_batches(dev):
    bb = PointNet2MSG(SMALL_CFG, input_channels=4).to(dev).eval()
    cl = synthetic.uniform_clouds(2, 1024, 1)
    pts = torch.from_numpy(synthetic.to_batch_points(cl)[:-5]).to(dev)  # second sample 5 points short
    with pytest.raises(AssertionError):
        bb({'batch_size': 2, 'points': pts})


def test_training_step_backward_runs(dev):
    """forward+backward through SA (fused grouping) + FP + neck: gradients reach every parameter."""
    torch.manual_seed(5)
    bb = PointNet2MSG(SMALL_CFG, input_channels=4).to(dev).train()
    neck = PDMNeck({'SOURCE_LAYER': 2, 'FEATURE_DIM': 16, 'DILATION': [3, 3, 1], 'SH_DEGREE': 1, 'BEV_STRIDE': 16,
                    'HEIGHT_BINS': 1, 'INPUT_CHANNELS': 256}, grid_size=[1408, 1600, 40],
                   voxel_size=[0.05, 0.05, 0.1], point_cloud_range=list(synthetic.KITTI_RANGE)).to(dev).train()
    cl = synthetic.lidar_like_clouds(2, 2048, 8)
    pts = torch.from_numpy(synthetic.to_batch_points(cl)).to(dev)
    bd = neck(bb({'batch_size': 2, 'points': pts}))
    loss = bd['point_features'].square().mean() + bd['spatial_features'].square().mean()
    loss.backward()
    for name, p in list(bb.named_parameters()) + list(neck.named_parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), name


@pytest.mark.parametrize("depth", [1, 2])
def test_pipelined_hot_path_equals_serial(dev, depth):
    """PipelinedHotPath (sampling hoisted one / two batches ahead on side streams) returns, for every batch of a
    sequence of DIFFERENT batches, exactly what backbone + neck return when run serially on that batch."""
    from pdm_ssd_amd import synthetic
    from pdm_ssd_amd.pdm_neck import PDMNeck
    from pdm_ssd_amd.pipeline import PipelinedHotPath
    from pdm_ssd_amd.pointnet2_backbone import PointNet2MSG
    torch.manual_seed(0)
    cfg = {'SA_CONFIG': {'NPOINTS': [256, 64], 'RADIUS': [[0.5, 1.0], [1.0, 2.0]], 'NSAMPLE': [[16, 32], [16, 32]],
                         'MLPS': [[[16, 16], [16, 32]], [[32, 32], [32, 64]]]}, 'FP_MLPS': [[32, 32], [64, 64]]}
    backbone = PointNet2MSG(cfg, input_channels=4).to(dev).eval()
    neck = PDMNeck({'SOURCE_LAYER': 1, 'FEATURE_DIM': 32, 'DILATION': [3, 3, 1], 'SH_DEGREE': 1, 'INPUT_CHANNELS': 48},
                   grid_size=[1408, 1600, 40], voxel_size=[0.05, 0.05, 0.1],
                   point_cloud_range=list(synthetic.KITTI_RANGE)).to(dev).eval()
    B, N = 2, 1024
    batches = [torch.from_numpy(synthetic.to_batch_points(synthetic.lidar_like_clouds(B, N, 100 + i))).to(dev) for i in range(5)]
    with torch.no_grad():
        ref = []
        for p in batches:
            bd = neck(backbone({'batch_size': B, 'points': p}))
            ref.append((bd['point_features'].clone(), bd['spatial_features'].clone()))
        pipe = PipelinedHotPath(backbone, neck, depth=depth)
        pipe.prime(batches[0], B, points_next=batches[1])
        for i in range(3):
            bd = pipe.step(batches[i], batches[i + 1], B, points_next2=batches[i + 2])
            torch.cuda.synchronize()
            assert torch.equal(bd['point_features'], ref[i][0]), f"batch {i}"
            assert torch.equal(bd['spatial_features'], ref[i][1]), f"batch {i}"


def test_pipelined_depth2_large_level1_uses_fresh_grids(dev):
    """depth 2 hands level 1's sampled set over in a STATIC buffer (pdm_copy_many rewrites it every step without
    touching the tensor's _version) and searches it with the grid kernels once it has >= 2048 points: every batch of a
    sequence of different batches must still equal the serial path (a search grid kept from the previous step would
    return the previous batch's neighbours)."""
    from pdm_ssd_amd.pipeline import PipelinedHotPath
    torch.manual_seed(0)
    cfg = {'SA_CONFIG': {'NPOINTS': [2048, 256], 'RADIUS': [[0.5, 1.0], [1.0, 2.0]], 'NSAMPLE': [[16, 32], [16, 32]],
                         'MLPS': [[[16, 16], [16, 32]], [[32, 32], [32, 64]]]}, 'FP_MLPS': [[32, 32], [64, 64]]}
    backbone = PointNet2MSG(cfg, input_channels=4).to(dev).eval()
    B, N = 2, 8192
    batches = [torch.from_numpy(synthetic.to_batch_points(synthetic.lidar_like_clouds(B, N, 500 + i))).to(dev) for i in range(8)]
    with torch.no_grad():
        ref = [backbone({'batch_size': B, 'points': p})['point_features'].clone() for p in batches[:6]]
        pipe = PipelinedHotPath(backbone, None, depth=2)
        pipe.prime(batches[0], B, points_next=batches[1])
        for i in range(6):
            bd = pipe.step(batches[i], batches[i + 1], B, points_next2=batches[i + 2])
            torch.cuda.synchronize()
            assert torch.equal(bd['point_features'], ref[i]), f"batch {i}"


@pytest.mark.parametrize("depth", [3, 4])
def test_segmented_fps_pipeline_equals_serial(dev, depth):
    """depth >= 3: the level-1 FPS is cut into depth - 1 resumable segments that run for different batches side by side;
    every batch of a sequence of DIFFERENT batches still comes out exactly as from the serial path."""
    from pdm_ssd_amd import synthetic
    from pdm_ssd_amd.pdm_neck import PDMNeck
    from pdm_ssd_amd.pipeline import PipelinedHotPath
    from pdm_ssd_amd.pointnet2_backbone import PointNet2MSG
    torch.manual_seed(0)
    cfg = {'SA_CONFIG': {'NPOINTS': [301, 64], 'RADIUS': [[0.5, 1.0], [1.0, 2.0]], 'NSAMPLE': [[16, 32], [16, 32]],
                         'MLPS': [[[16, 16], [16, 32]], [[32, 32], [32, 64]]]}, 'FP_MLPS': [[32, 32], [64, 64]]}
    backbone = PointNet2MSG(cfg, input_channels=4).to(dev).eval()
    neck = PDMNeck({'SOURCE_LAYER': 1, 'FEATURE_DIM': 32, 'DILATION': [3, 3, 1], 'SH_DEGREE': 1, 'INPUT_CHANNELS': 48},
                   grid_size=[1408, 1600, 40], voxel_size=[0.05, 0.05, 0.1],
                   point_cloud_range=list(synthetic.KITTI_RANGE)).to(dev).eval()
    B, N, S = 2, 2048, depth - 1
    nb = 4 + S + 1
    batches = [torch.from_numpy(synthetic.to_batch_points(synthetic.lidar_like_clouds(B, N, 300 + i))).to(dev) for i in range(nb)]
    with torch.no_grad():
        ref = []
        for p in batches[:4]:
            bd = neck(backbone({'batch_size': B, 'points': p}))
            ref.append((bd['point_features'].clone(), bd['spatial_features'].clone()))
        pipe = PipelinedHotPath(backbone, neck, depth=depth)
        pipe.prime_segmented(batches[:S + 1], B)
        for i in range(4):
            bd = pipe.step(batches[i], None, B, points_ahead=batches[i + 1:i + 2 + S])
            torch.cuda.synchronize()
            assert torch.equal(bd['point_features'], ref[i][0]), f"batch {i}"
            assert torch.equal(bd['spatial_features'], ref[i][1]), f"batch {i}"


def test_autotune_hoisting(dev):
    """Measured choice between the hoisted and the plain first SA layer per level (which one wins depends on how many
    rows the compacted neighbour lists hold); the features agree to 1e-4 either way (fp32 summation order differs)."""
    from pdm_ssd_amd import synthetic
    from pdm_ssd_amd.pointnet2_backbone import PointNet2MSG
    torch.manual_seed(1)
    cfg = {'SA_CONFIG': {'NPOINTS': [512, 128], 'RADIUS': [[0.2, 0.4], [0.4, 0.8]], 'NSAMPLE': [[16, 32], [16, 32]],
                         'MLPS': [[[16, 16, 32], [16, 16, 32]], [[64, 64], [64, 96]]]}, 'FP_MLPS': [[64, 64], [128, 128]]}
    net = PointNet2MSG(cfg, input_channels=4).to(dev).eval()
    B, N = 2, 4096
    sparse = torch.from_numpy(synthetic.to_batch_points(synthetic.uniform_clouds(B, N, 7))).to(dev)
    with torch.no_grad():
        before = net({'batch_size': B, 'points': sparse})['point_features'].clone()
        dec = net.autotune_hoisting(sparse, B)
        assert len(dec) == 2 and dec[0]['use_pre'] is False and dec[0]['ms_hoisted'] is None   # one input feature: never hoisted
        assert dec[1]['ms_hoisted'] > 0 and dec[1]['ms_plain'] > 0
        assert dec[1]['use_pre'] == (dec[1]['ms_hoisted'] < dec[1]['ms_plain']) == net.SA_modules[1].use_pre
        after = net({'batch_size': B, 'points': sparse})['point_features']
        torch.testing.assert_close(after, before, rtol=1e-4, atol=1e-4)
        for form in (True, False):   # either form of the wide level gives the same features to 1e-4
            net.SA_modules[1].use_pre = form
            out = net({'batch_size': B, 'points': sparse})['point_features']
            torch.testing.assert_close(out, before, rtol=1e-4, atol=1e-4)


def test_full_size_paths_agree(dev):
    """BASELINE's network at its real point count (PointNet2MSG pointrcnn config + PDM neck, 16384-point clouds): the
    pipelined step (compacted neighbour lists, level-1 FPS in two resumable segments, three batches in flight) returns
    bit for bit what the serial path returns, and the compacted SA kernels bit for bit what the dense ones return —
    size-independent properties at the size the bench runs."""
    import bench
    from pdm_ssd_amd import synthetic
    from pdm_ssd_amd.pipeline import PipelinedHotPath
    backbone, neck = bench.build_models(dev)
    B, N = 4, 16384
    batches = [torch.from_numpy(synthetic.to_batch_points(
        (synthetic.lidar_like_clouds if i % 2 else synthetic.uniform_clouds)(B, N, 900 + 10 * i))).to(dev) for i in range(5)]
    with torch.no_grad():
        ref = []
        for p in batches[:2]:
            bd = neck(backbone({'batch_size': B, 'points': p}))
            ref.append((bd['point_features'].clone(), bd['spatial_features'].clone()))
        pipe = PipelinedHotPath(backbone, neck, depth=3)
        pipe.prime_segmented(batches[:3], B)
        for i in range(2):
            bd = pipe.step(batches[i], None, B, points_ahead=batches[i + 1:i + 4])
            torch.cuda.synchronize()
            assert torch.equal(bd['point_features'], ref[i][0]) and torch.equal(bd['spatial_features'], ref[i][1]), f"batch {i}"
        for sa in backbone.SA_modules:
            sa.use_pack = False
        bd = neck(backbone({'batch_size': B, 'points': batches[1]}))
        assert torch.equal(bd['point_features'], ref[1][0]) and torch.equal(bd['spatial_features'], ref[1][1])


@pytest.mark.parametrize("bf16", [False, True])
def test_query_and_group_channels_last(dev, bf16):
    """The training path's grouped tensor born in channels-last memory (bf16 under autocast) equals the reference-layout
    tensor (cast to bf16 by round-to-nearest-even) bit for bit, and its gradient equals the reference-layout gradient."""
    from pdm_ssd_amd import synthetic
    from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
    cl = synthetic.lidar_like_clouds(2, 2048, 3)
    xyz = torch.from_numpy(np.ascontiguousarray(cl[:, :, :3])).to(dev)
    new_xyz = xyz[:, :300].contiguous()
    feats = torch.randn(2, 37, 2048, device=dev)
    f1 = feats.clone().requires_grad_(True)
    f2 = feats.clone().requires_grad_(True)
    ref_mod = pu.QueryAndGroup(0.9, 16)
    cl_mod = pu.QueryAndGroup(0.9, 16)
    cl_mod.channels_last = True
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
        want = ref_mod(xyz, new_xyz, f1)
        got = cl_mod(xyz, new_xyz, f2)
    assert got.shape == want.shape == (2, 40, 300, 16)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert got.dtype == (torch.bfloat16 if bf16 else torch.float32)
    assert torch.equal(got.float(), want.to(got.dtype).float())
    go = torch.randn_like(want)
    want.backward(go)
    got.backward(go.to(got.dtype).contiguous(memory_format=torch.channels_last))
    tol = 2e-2 if bf16 else 1e-5
    torch.testing.assert_close(f2.grad, f1.grad, rtol=tol, atol=tol)


@pytest.mark.parametrize("c,rows_feat", [(37, False), (5, True), (96, True), (0, False)])
def test_query_and_group_rows_form_zero_padded(dev, oracle, c, rows_feat):
    """The rows form of the training path (pad_to_8): (B, round8(3 + C), M, ns) bf16, the first 3 + C channels equal the
    reference tensor rounded to nearest even (QueryAndGroup of the oracle), the padding channels are exact zeros, features may
    arrive as a (B, C, N) VIEW of point-major storage; the gradient reads only the real channels."""
    from pdm_ssd_amd import synthetic
    from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
    cl = synthetic.lidar_like_clouds(2, 2048, 3)
    xyz_np = np.ascontiguousarray(cl[:, :, :3])
    xyz = torch.from_numpy(xyz_np).to(dev)
    new_xyz = xyz[:, :300].contiguous()
    g = torch.Generator().manual_seed(c)
    f_np = torch.randn(2, c, 2048, generator=g).numpy() if c else None
    if c:
        f = torch.from_numpy(f_np).to(dev)
        feats = (f.transpose(1, 2).contiguous().transpose(1, 2) if rows_feat else f).requires_grad_(True)
    else:
        feats = None
    mod = pu.QueryAndGroup(0.9, 16)
    mod.channels_last, mod.pad_to_8 = True, True
    with torch.autocast('cuda', dtype=torch.bfloat16):
        got = mod(xyz, new_xyz, feats)
    cp = (3 + c + 7) // 8 * 8
    assert got.shape == (2, cp, 300, 16) and got.dtype == torch.bfloat16 and got.permute(0, 2, 3, 1).is_contiguous()
    want, _ = oracle.query_and_group(0.9, 16, xyz_np, xyz_np[:, :300].copy(), f_np)
    assert torch.equal(got[:, :3 + c].float().cpu(), torch.from_numpy(want).bfloat16().float())
    assert float(got[:, 3 + c:].float().abs().max()) == 0.0 if cp > 3 + c else True
    if c:
        go = torch.randn(2, cp, 300, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        got.backward(go)
        idx = oracle.ball_query(0.9, 16, xyz_np, xyz_np[:, :300].copy())
        ref = oracle.grouping_operation_grad(np.ascontiguousarray(go[:, 3:3 + c].float().cpu().numpy()), idx, 2048)
        np.testing.assert_allclose(feats.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("c2,c1,kb,sb", [(256, 1, True, False), (64, 96, True, True), (40, 0, False, False), (12, 5, False, True)])
def test_interp_concat_rows_matches_oracle(dev, oracle, c2, c1, kb, sb):
    """InterpConcatRows = cat([three_interpolate(known, idx, w), skip], 1) written as zero-padded bf16 rows: every element the
    oracle's fp32 value (pinned fma order) rounded to nearest even — BIT-exact; the gradient towards the known features
    against the oracle's scatter-add (fp32 order: 1e-4), the skip gradient = the column block of the incoming gradient."""
    from pdm_ssd_amd import synthetic
    from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
    B, n, m = 2, 1500, 400
    cl = synthetic.lidar_like_clouds(B, n, 8)
    unknown = np.ascontiguousarray(cl[:, :, :3]); known = np.ascontiguousarray(cl[:, :m, :3]) + np.float32(0.05)
    dist, idx = oracle.three_nn(unknown, known)
    w = 1.0 / (dist + 1e-8); w = (w / w.sum(2, keepdims=True)).astype(np.float32)
    g = torch.Generator().manual_seed(c2 + c1)
    kf = torch.randn(B, m, c2, generator=g)
    sf = torch.randn(B, n, c1, generator=g) if c1 else None
    if kb: kf = kf.bfloat16().float()
    if sb and c1: sf = sf.bfloat16().float()
    known_rows = (kf.bfloat16() if kb else kf).to(dev).requires_grad_(True)
    skip_rows = None if sf is None else (sf.bfloat16() if sb else sf).to(dev).requires_grad_(True)
    out = pu.interp_concat_rows(known_rows, skip_rows, torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev))
    ld = (c2 + c1 + 7) // 8 * 8
    assert out.shape == (B, ld, n, 1) and out.dtype == torch.bfloat16 and out.permute(0, 2, 3, 1).is_contiguous()
    interp = oracle.three_interpolate(np.ascontiguousarray(kf.numpy().transpose(0, 2, 1)), idx, w)      # (B, c2, n)
    want = torch.from_numpy(interp) if sf is None else torch.cat([torch.from_numpy(interp), sf.permute(0, 2, 1)], 1)
    assert torch.equal(out[:, :c2 + c1, :, 0].float().cpu(), want.bfloat16().float())
    if ld > c2 + c1:
        assert float(out[:, c2 + c1:].float().abs().max()) == 0.0
    go = torch.randn(B, ld, n, 1, generator=g).bfloat16().to(dev)
    out.backward(go)
    ref = oracle.three_interpolate_grad(np.ascontiguousarray(go[:, :c2, :, 0].float().cpu().numpy()), idx, w, m)   # (B, c2, m)
    tol = 2e-2 if kb else 1e-4        # a bf16 known tensor receives its gradient rounded to bf16
    np.testing.assert_allclose(known_rows.grad.float().cpu().numpy(), ref.transpose(0, 2, 1), rtol=tol, atol=tol)
    if c1:
        assert torch.equal(skip_rows.grad.float().cpu(), go[:, c2:c2 + c1, :, 0].float().cpu().permute(0, 2, 1))
