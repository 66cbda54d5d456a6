"""fp32 emulated on the bf16 matrix pipe (csrc/rows_chain_x3.hip, OPT-IN): every fp32 operand as three bf16 pieces, six
partial products per product, fp32 accumulation.  The point head's stacks (make_fc_layers of
/root/reference/pcdet/models/dense_heads/point_head_template.py:35-48, eval mode) through it against the fp32-MFMA chain
kernel (the default path) and against the same stack in float64 on the CPU.

Bounds stated here: |x3 - fp32 MFMA| <= 4e-6 of the output's scale (two different fp32 roundings of the same sums; the
dropped partial products are <= 3 * 2^-24 of each product), both within 1e-4 (north_star's feature tolerance; measured
~1e-6) of float64."""
import numpy as np
import pytest
import torch

from pdm_ssd_amd import fused
from pdm_ssd_amd.dense_heads.point_head_box import _fc_layers
from pdm_ssd_amd.dense_heads.point_head_template import PointHeadTemplate


def test_split_is_exact_and_pieces_are_bf16():
    rng = np.random.default_rng(0)
    w = np.concatenate([rng.standard_normal(100000).astype(np.float32) * 10.0 ** rng.integers(-6, 6, 100000),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.0e38, 1.2e-30, 1 + 2.0 ** -23, 1 - 2.0 ** -24], dtype=np.float32)])
    hi, mid, lo = fused.split_bf16x3(w)
    for p in (hi, mid, lo):
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0)                  # representable in bfloat16
    rec = (hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64))
    err = np.abs(rec - w.astype(np.float64))
    normal = np.abs(w) >= 2.0 ** -100                                     # (pieces of values near the subnormal range underflow)
    # 8 + 8 + 8 significand bits with round-to-nearest pieces: a residual may need a 9th bit, so the sum is within half an
    # fp32 ulp of the value (2^-24 relative), not always equal to it — the same order as the dropped partial products
    assert np.all(err[normal] <= np.abs(w[normal].astype(np.float64)) * 2.0 ** -24)
    assert float((err == 0).mean()) > 0.1
    assert np.all(np.abs(mid) <= np.abs(hi) * 2.0 ** -7 + 1e-45) and np.all(np.abs(lo) <= np.abs(hi) * 2.0 ** -15 + 1e-45)


def test_fragment_layout_round_trips():
    """_x3_fragments: lane (kg, row) element j of fragment (mb, kb) is W[16 mb + row][32 kb + 16 (j >> 2) + 4 kg + (j & 3)]."""
    rng = np.random.default_rng(1)
    w = fused._bf16_rne(rng.standard_normal((32, 64)).astype(np.float32))     # bf16-exact: the hi piece is the value itself
    fr = fused._x3_fragments(w)
    assert fr.shape == (3, 2, 2, 64, 8) and not fr[1].any() and not fr[2].any()
    bits = (w.view(np.uint32) >> 16).astype(np.uint16)
    for mb, kb, lane, j in [(0, 0, 0, 0), (1, 1, 63, 7), (0, 1, 17, 5), (1, 0, 40, 2)]:
        row, kg = lane & 15, lane >> 4
        assert fr[0, mb, kb, lane, j] == bits[16 * mb + row, 32 * kb + 16 * (j >> 2) + 4 * kg + (j & 3)]


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cout", [(8192, 3), (8192 + 29, 8), (3 * 16384, 8)])
def test_x3_chain_matches_fp32_mfma_and_float64(dev, rows, cout):
    torch.manual_seed(rows + cout)
    seq = PointHeadTemplate.make_fc_layers([256, 256], 128, cout).eval()
    for m in seq.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    x = torch.randn(rows, 128)
    with torch.no_grad():
        want = seq.double()(x.double())
    seq = seq.float()
    layers = _fc_layers(seq)
    pk32, pkx3 = fused.PackedMLP(layers, dev), fused.PackedMLPx3(layers, dev)
    xg = x.to(dev)
    stride = (cout + 3) // 4 * 4
    a = torch.full((rows, stride), float('nan'), device=dev)
    b = torch.full((rows, stride), float('nan'), device=dev)
    fused.rows_forward(pk32, xg, a, relu_last=False)
    fused.rows_forward_x3(pkx3, xg, b, relu_last=False)
    torch.cuda.synchronize()
    a, b = a[:, :cout].cpu().double(), b[:, :cout].cpu().double()
    scale = float(want.abs().max())
    assert torch.isfinite(b).all()
    e_ab = float((a - b).abs().max()) / scale
    e_a, e_b = float((a - want).abs().max()) / scale, float((b - want).abs().max()) / scale
    print(f"rows={rows} cout={cout}: |x3 - fp32 MFMA| {e_ab:.2e}, fp32 MFMA vs f64 {e_a:.2e}, x3 vs f64 {e_b:.2e} (of the output scale {scale:.2f})")
    assert e_ab <= 4e-6 and e_a <= 1e-4 and e_b <= 1e-4
    assert e_b <= 4 * max(e_a, 5e-7)          # the emulation is as accurate as the fp32 matrix instruction


@pytest.mark.gpu
def test_point_head_opt_in_switch(dev):
    """PointHeadBox.use_x3 = True routes both stacks through the emulation; the decoded boxes and scores agree with the
    default path to 1e-5 (the class arg-max may flip only where two logits tie to that accuracy)."""
    from pdm_ssd_amd import _native
    from pdm_ssd_amd.detector_config import build_pdm_ssd
    torch.manual_seed(0)
    head = build_pdm_ssd().point_head.to(dev).eval()
    n = 16384
    bd = {'batch_size': 1, 'point_features': torch.randn(n, 128, device=dev),
          'point_coords': torch.cat([torch.zeros(n, 1, device=dev), torch.rand(n, 3, device=dev) * 40], dim=1)}
    calls = []
    orig = _native.call
    _native.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        with torch.no_grad():
            ref = head(dict(bd))
            assert "pdm_rows_mlp_x3" not in calls
            head.use_x3 = True
            got = head(dict(bd))
            assert calls.count("pdm_rows_mlp_x3") == 2
    finally:
        _native.call = orig
    torch.testing.assert_close(got['batch_cls_preds'], ref['batch_cls_preds'], rtol=1e-5, atol=1e-5)
    top2 = torch.topk(ref['batch_cls_preds'], 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    torch.testing.assert_close(got['batch_box_preds'][clear], ref['batch_box_preds'][clear], rtol=1e-4, atol=1e-4)
