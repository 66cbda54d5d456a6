"""Host-side logic that needs no GPU: block-size rule, weight folding/packing, grid parameters,
synthetic data, config bookkeeping, sharding."""
import copy
import math

import numpy as np
import pytest
import torch

from oracle import cpu_oracle as o
from pdm_ssd_amd import dist_utils, fused, pdm_ops, synthetic


def test_block_size_rule_integer_form_equals_reference_double_form():
    # sampling.hip uses integer floor(log2 n); the reference truncates log(n)/log(2.0) (cuda_utils.h:11)
    for n in list(range(1, 70000)) + [2 ** k + d for k in range(16, 21) for d in (-1, 0, 1)]:
        ilog = n.bit_length() - 1
        assert o.opt_n_threads(n) == min(1 << ilog, 1024), n
        assert int(math.log(n) / math.log(2.0)) == ilog, n


def test_fold_conv_bn_equals_eval_mode_sequence():
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(7, 5, 1, bias=False)
    bn = torch.nn.BatchNorm2d(5).eval()
    with torch.no_grad():
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2); bn.weight.normal_(); bn.bias.normal_()
    x = torch.randn(3, 7, 4, 2)
    w, shift = fused.fold_conv_bn(conv, bn)
    got = torch.relu(torch.einsum('oc,bchw->bohw', w.float(), x) + shift.float()[None, :, None, None])
    torch.testing.assert_close(got, torch.relu(bn(conv(x))), rtol=1e-5, atol=1e-5)


def test_pack_layer_layout_matches_header_contract():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((40, 35)).astype(np.float32)
    b = rng.standard_normal(40).astype(np.float32)
    pw, pb, kp, cp = fused.pack_layer(w, b)
    assert (kp, cp) == (48, 48) and pw.size == kp * cp
    P = pw.reshape(cp // 16, kp // 16, 64, 4)
    for mb in range(cp // 16):
        for kb in range(kp // 16):
            for lane in (0, 17, 37, 63):
                for s in range(4):
                    oc, k = 16 * mb + (lane & 15), 16 * kb + 4 * (lane >> 4) + s
                    assert P[mb, kb, lane, s] == (w[oc, k] if oc < 40 and k < 35 else 0.0)
    assert np.array_equal(pb[:40], b) and (pb[40:] == 0).all()


def test_split_shared_mlp_rejects_other_patterns():
    good = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 1, bias=False), torch.nn.BatchNorm2d(8), torch.nn.ReLU())
    assert len(fused.split_shared_mlp(good)) == 1
    bad = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, bias=False), torch.nn.BatchNorm2d(8), torch.nn.ReLU())
    assert fused.split_shared_mlp(bad) is None
    assert fused.split_shared_mlp(torch.nn.Sequential(torch.nn.Conv2d(4, 8, 1), torch.nn.ReLU())) is None


def test_bev_grid_params_are_bitwise_the_oracles():
    rng_, cell = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0), (0.4, 0.4, 4.0)
    g = pdm_ops.BevGrid(rng_, cell)
    origin, c, ic, dims = o.pdm_grid_params(rng_, cell)
    assert (g.W, g.H, g.D) == dims == (176, 200, 1)
    assert g.origin.tobytes() == origin.tobytes() and g.cell.tobytes() == c.tobytes() and g.inv_cell.tobytes() == ic.tobytes()


def test_synthetic_clouds_are_deterministic_and_in_range():
    a, b = synthetic.uniform_clouds(3, 500), synthetic.uniform_clouds(3, 500)
    assert np.array_equal(a, b) and a.dtype == np.float32 and a.shape == (3, 500, 4)
    assert not np.array_equal(a[0], a[1])  # seed = 1234 + sample index
    l = synthetic.lidar_like_clouds(2, 2000)
    for arr in (a, l):
        assert (arr[..., 0] >= 0).all() and (arr[..., 0] < 70.4).all() and (np.abs(arr[..., 1]) <= 40).all()
        assert (arr[..., 2] >= -3).all() and (arr[..., 2] < 1).all()
    pts = synthetic.to_batch_points(a)
    assert pts.shape == (1500, 5) and (pts[:500, 0] == 0).all() and (pts[1000:, 0] == 2).all()


def test_sa_constructor_mutates_spec_like_the_reference():
    from pdm_ssd_amd.pointnet2_batch import pointnet2_modules as pm
    spec = [[4, 8, 16]]
    pm.PointnetSAModuleMSG(npoint=8, radii=[1.0], nsamples=[4], mlps=spec)
    assert spec[0][0] == 7  # ref pointnet2_modules.py:86-88
    sa = pm.PointnetSAModule(mlp=[2, 4], npoint=4, radius=0.5, nsample=2)
    assert sa.mlps[0][0].weight.shape == (4, 5, 1, 1)


def test_backbone_channel_bookkeeping():
    from pdm_ssd_amd.pointnet2_backbone import POINTRCNN_MSG_CFG, PointNet2MSG
    cfg = copy.deepcopy(POINTRCNN_MSG_CFG)
    bb = PointNet2MSG(cfg, input_channels=4)
    assert bb.num_point_features == 128 and cfg == POINTRCNN_MSG_CFG  # caller's config untouched
    assert [m.mlp[0].in_channels for m in bb.FP_modules] == [257, 608, 768, 1536]
    assert [sum(s[-2].num_features for s in sa.mlps) for sa in bb.SA_modules] == [96, 256, 512, 1024]


@pytest.mark.parametrize("total,world", [(256, 8), (32, 3), (5, 8), (0, 2)])
def test_shard_range_partitions_whole_clouds(total, world):
    spans = [dist_utils.shard_range(total, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [e - b for b, e in spans]
    assert max(sizes) - min(sizes) <= 1


def test_train_sequential_after_convert_sync_batchnorm():
    """tools/train.py:130-131 of the reference turns every BatchNorm into SyncBatchNorm under --sync_bn.  TrainSequential keeps
    its child indices (state_dict keys) through the conversion, still pairs (BatchNorm, ReLU) for its fused operators while
    there is nothing to synchronise with, and hands a multi-rank SyncBatchNorm back to torch."""
    import torch
    from pdm_ssd_amd import fused_bn
    seq = fused_bn.TrainSequential(torch.nn.Conv2d(8, 16, 1, bias=False), torch.nn.BatchNorm2d(16), torch.nn.ReLU(),
                                   torch.nn.Conv2d(16, 8, 1, bias=False), torch.nn.BatchNorm2d(8), torch.nn.ReLU())
    keys = list(seq.state_dict())
    conv = torch.nn.SyncBatchNorm.convert_sync_batchnorm(seq)
    assert type(conv) is fused_bn.TrainSequential and list(conv.state_dict()) == keys
    assert isinstance(conv[1], torch.nn.SyncBatchNorm) and isinstance(conv[1], fused_bn._BN)      # no process group: a local BatchNorm
    assert isinstance(torch.nn.BatchNorm1d(4), fused_bn._BN) and not isinstance(torch.nn.ReLU(), fused_bn._BN)
    want, pad = fused_bn._stats_wanted(list(conv), 0)
    assert want and not pad
    x = torch.randn(2, 8, 5, 3)
    y = conv.train()(x)                                       # CPU: the torch modules (SyncBatchNorm without a group = batch_norm)
    ref = seq.train()(x)
    torch.testing.assert_close(y, ref)

    class FakeGroup:                                          # what a two-rank group looks like to the check
        pass
    import torch.distributed as dist
    real = (dist.is_initialized, dist.get_world_size)
    dist.is_initialized, dist.get_world_size = (lambda: True), (lambda group=None: 2)
    try:
        assert not isinstance(conv[1], fused_bn._BN)
        assert fused_bn._stats_wanted(list(conv), 0) == (False, False)
    finally:
        dist.is_initialized, dist.get_world_size = real
