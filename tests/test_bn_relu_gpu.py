"""Fused training-mode BatchNorm + ReLU (csrc/bn_relu.hip, pdm_ssd_amd/fused_bn.py) against torch's own
batch_norm + relu in fp32 on the CPU: outputs, running statistics, gradients of the input and of gamma / beta.
fp32 activations 1e-5; bf16 activations within bf16 rounding of the fp32 result computed from the same bf16 inputs."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from pdm_ssd_amd import fused_bn


def reference(x, bn_cpu, relu, gy):
    x = x.detach().float().cpu().requires_grad_(True)
    y = bn_cpu(x)
    if relu:
        y = torch.relu(y)
    y.backward(gy.float().cpu())
    return y.detach(), x.grad, bn_cpu.weight.grad, bn_cpu.bias.grad


CASES = [
    # shape, channels_last, dtype, relu
    ((5000, 64), False, torch.float32, True),          # Linear output (rows, C)
    ((5000, 64), False, torch.bfloat16, True),
    ((2, 32, 50, 16), True, torch.float32, True),      # SA group tensor, channels-last
    ((2, 96, 37, 8), True, torch.bfloat16, True),      # C / 8 = 12 threads per row: does not divide 256
    ((3, 24, 1024), False, torch.float32, True),       # Conv1d output (B, C, L)
    ((2, 16, 256, 1), False, torch.bfloat16, True),    # FP module: (B, C, n, 1), position fastest
    ((4, 128, 512, 1), False, torch.float32, False),   # BatchNorm without ReLU
    ((70000, 256), False, torch.bfloat16, True),       # more rows than one pass of the grid
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cl,dtype,relu", CASES)
def test_fused_bn_relu_matches_torch(dev, shape, cl, dtype, relu):
    torch.manual_seed(sum(shape))
    C = shape[1]
    bn = (nn.BatchNorm1d if len(shape) <= 3 else nn.BatchNorm2d)(C, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5); bn.bias.copy_(torch.randn(C) * 0.2)
        bn.running_mean.copy_(torch.randn(C) * 0.1); bn.running_var.copy_(torch.rand(C) + 0.5)
    bn_g = copy.deepcopy(bn).to(dev).train()
    x = (torch.randn(shape) * 1.5 + 0.3).to(dtype)
    gy = torch.randn(shape).to(dtype)
    xg = x.to(dev)
    if cl:
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_(True)
    assert fused_bn.applies(xg, bn_g)
    y = fused_bn.batch_norm_relu(xg, bn_g, relu)
    assert y.dtype == dtype and y.stride() == xg.stride()
    y.backward(gy.to(dev))
    ry, rdx, rdw, rdb = reference(x, bn.train(), relu, gy)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(y.detach().float().cpu(), ry, **tol)
    torch.testing.assert_close(xg.grad.float().cpu(), rdx, **(tol if dtype == torch.float32 else dict(rtol=5e-2, atol=2e-2)))
    scale = max(1.0, float(rdw.abs().max()))
    torch.testing.assert_close(bn_g.weight.grad.cpu(), rdw, rtol=1e-4 if dtype == torch.float32 else 2e-2, atol=(1e-4 if dtype == torch.float32 else 2e-2) * scale)
    torch.testing.assert_close(bn_g.bias.grad.cpu(), rdb, rtol=1e-4 if dtype == torch.float32 else 2e-2, atol=(1e-4 if dtype == torch.float32 else 2e-2) * scale)
    torch.testing.assert_close(bn_g.running_mean.cpu(), bn.running_mean, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bn_g.running_var.cpu(), bn.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn_g.num_batches_tracked) == 1


@pytest.mark.gpu
def test_train_sequential_fuses_and_falls_back(dev):
    """Conv -> BN -> ReLU -> Conv -> BN -> ReLU through TrainSequential == through nn.Sequential (fp32); shapes the
    kernels do not take (C = 6) and eval mode go through torch."""
    torch.manual_seed(1)
    layers = [nn.Conv2d(8, 6, 1, bias=False), nn.BatchNorm2d(6), nn.ReLU(), nn.Conv2d(6, 32, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU()]
    plain = nn.Sequential(*copy.deepcopy(layers)).to(dev).train()
    fused = fused_bn.TrainSequential(*copy.deepcopy(layers)).to(dev).train()
    assert list(plain.state_dict()) == list(fused.state_dict())
    x = torch.randn(4, 8, 64, 16, device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = plain(xa), fused(xb)
    torch.testing.assert_close(ya, yb, rtol=1e-4, atol=1e-5)
    ya.square().sum().backward(); yb.square().sum().backward()
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-3, atol=1e-4)
    for (k, pa), (_, pb) in zip(plain.named_parameters(), fused.named_parameters()):
        torch.testing.assert_close(pa.grad, pb.grad, rtol=1e-3, atol=1e-3, msg=k)
    for (k, ba), (_, bb) in zip(plain.named_buffers(), fused.named_buffers()):
        torch.testing.assert_close(ba.float(), bb.float(), rtol=1e-4, atol=1e-5, msg=k)
    plain.eval(); fused.eval()
    with torch.no_grad():
        torch.testing.assert_close(plain(x), fused(x))


def test_train_sequential_is_a_plain_sequential_on_cpu():
    seq = fused_bn.TrainSequential(nn.Linear(4, 8, bias=False), nn.BatchNorm1d(8), nn.ReLU()).train()
    ref = nn.Sequential(*copy.deepcopy(list(seq))).train()
    x = torch.randn(16, 4)
    torch.testing.assert_close(seq(x), ref(x))
    assert isinstance(seq, nn.Sequential) and list(seq.state_dict()) == list(ref.state_dict())


@pytest.mark.gpu
def test_tall_linear_split_k_weight_gradient(dev):
    """Linear over 65536+ rows under bf16 autocast: same output as nn.Linear, weight / bias / input gradients equal to
    the fp32 result within bf16 rounding (the split-K sum is carried in fp32)."""
    torch.manual_seed(3)
    lin = nn.Linear(64, 48, bias=True).to(dev)
    x = torch.randn(8 * 8192, 64, device=dev)
    gy = torch.randn(8 * 8192, 48, device=dev)
    xa = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = fused_bn.tall_linear(xa, lin)
        yb = lin(x)
    assert ya.dtype == torch.bfloat16 and torch.equal(ya, yb)
    ya.backward(gy.to(ya.dtype))
    ref_w = (gy.bfloat16().float().t() @ x.bfloat16().float())
    ref_b = gy.bfloat16().float().sum(0)
    ref_x = gy.bfloat16().float() @ lin.weight.detach().bfloat16().float()
    torch.testing.assert_close(lin.weight.grad, ref_w, rtol=1e-2, atol=1e-2 * float(ref_w.abs().max()))
    torch.testing.assert_close(lin.bias.grad, ref_b, rtol=1e-2, atol=1e-2 * float(ref_b.abs().max()))
    torch.testing.assert_close(xa.grad, ref_x, rtol=2e-2, atol=2e-2)
    # short matrices and fp32 runs go through the module itself
    assert fused_bn.tall_linear(x[:100], lin).dtype == torch.float32


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cl", [((8, 32, 1024, 16), True), ((8, 64, 4096, 1), False), ((4, 24, 300, 7), True)])
def test_conv1x1_split_k_weight_gradient(dev, shape, cl):
    """1x1 convolution under bf16 autocast through fused_bn.conv1x1: output equal to the module's within bf16 rounding (and in the same memory format), weight and input
    gradients equal to the fp32 result of the bf16-rounded operands within bf16 rounding."""
    torch.manual_seed(5)
    conv = nn.Conv2d(shape[1], 48, 1, bias=False).to(dev)
    x = torch.randn(shape, device=dev)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    xa = x.clone().requires_grad_(True)
    gy = torch.randn(shape[0], 48, shape[2], shape[3], device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = fused_bn.conv1x1(xa, conv)
        yb = conv(x)
    assert ya.dtype == torch.bfloat16 and ya.stride() == yb.stride()
    torch.testing.assert_close(ya.float(), yb.float(), rtol=1e-2, atol=1e-2)     # same bf16 GEMM, another kernel's summation order
    ya.backward(gy.to(ya.dtype))
    xr = x.bfloat16().float(); gr = gy.bfloat16().float(); wr = conv.weight.detach().bfloat16().float().view(48, -1)
    ref_w = torch.einsum("bohw,bihw->oi", gr, xr)
    ref_x = torch.einsum("bohw,oi->bihw", gr, wr)
    torch.testing.assert_close(conv.weight.grad.view(48, -1), ref_w, rtol=1e-2, atol=1e-2 * float(ref_w.abs().max()))
    torch.testing.assert_close(xa.grad, ref_x, rtol=2e-2, atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((2, 32, 200, 16), torch.float32), ((3, 64, 500, 32), torch.bfloat16), ((2, 96, 64, 7), torch.bfloat16)])
def test_bn_relu_max_pool_as_one_operator(dev, shape, dtype):
    """The tail of an SA scale in training: relu(bn(x)) max-pooled over nsample, one operator against torch's three on
    the CPU in fp32 — with duplicated neighbours (exact ties, as ball-query padding makes them), negative gammas (the
    pooled value then comes from the group's MINIMUM), groups that never pass the ReLU."""
    torch.manual_seed(shape[2])
    B, C, M, ns = shape
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C)); bn.bias.copy_(torch.randn(C) * 0.5)
        bn.weight[1] = 0.7; bn.bias[1] = -50.0                                       # a channel that is never positive
    x = (torch.randn(B, C, M, ns) * 1.3 + 0.2).to(dtype)
    x[..., ns // 2:] = x[..., :1]                                                     # padding copies of the first neighbour
    gy = torch.randn(B, C, M, 1).to(dtype)
    seq = fused_bn.TrainSequential(nn.Identity(), copy.deepcopy(bn), nn.ReLU()).to(dev).train()
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert fused_bn.pool_applies(xg, seq[1])
    y = seq.forward_max_pooled(xg)
    assert y.shape == (B, C, M, 1) and y.dtype == dtype
    y.backward(gy.to(dev))
    xc = x.float().requires_grad_(True)
    bn.train()
    yc = torch.nn.functional.max_pool2d(torch.relu(bn(xc)), kernel_size=[1, ns])
    yc.backward(gy.float())
    f32 = dtype == torch.float32
    torch.testing.assert_close(y.detach().float().cpu(), yc.detach(), rtol=1e-5 if f32 else 2e-2, atol=1e-5 if f32 else 2e-2)
    torch.testing.assert_close(xg.grad.float().cpu(), xc.grad, rtol=1e-4 if f32 else 5e-2, atol=1e-5 if f32 else 2e-2)
    sw, sb = max(1.0, float(bn.weight.grad.abs().max())), max(1.0, float(bn.bias.grad.abs().max()))
    torch.testing.assert_close(seq[1].weight.grad.cpu(), bn.weight.grad, rtol=1e-4 if f32 else 2e-2, atol=(1e-4 if f32 else 2e-2) * sw)
    torch.testing.assert_close(seq[1].bias.grad.cpu(), bn.bias.grad, rtol=1e-4 if f32 else 2e-2, atol=(1e-4 if f32 else 2e-2) * sb)
    torch.testing.assert_close(seq[1].running_mean.cpu(), bn.running_mean, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(seq[1].running_var.cpu(), bn.running_var, rtol=1e-4, atol=1e-5)
    assert float(y[:, 1].abs().max()) == 0.0 and float(xg.grad[:, 1].abs().max()) < 1e-3


@pytest.mark.gpu
def test_bn_relu_inside_the_next_contraction_is_bit_identical(dev):
    """Conv -> BN -> ReLU -> Conv under bf16 autocast: with the BatchNorm + ReLU applied in the second contraction's load path
    (fused_bn._BnReluRowsGemm: the normalised tensor is never written) outputs, input gradient and every parameter
    gradient are BIT-identical to the path that runs BatchNorm + ReLU as an operator of its own; odd widths (20 -> padded to
    24) and a biased last layer included; running statistics equal."""
    import copy
    from pdm_ssd_amd import fused_bn
    torch.manual_seed(5)
    net = fused_bn.TrainSequential(torch.nn.Conv2d(16, 32, 1, bias=False), torch.nn.BatchNorm2d(32), torch.nn.ReLU(),
                                   torch.nn.Conv2d(32, 20, 1, bias=False), torch.nn.BatchNorm2d(20), torch.nn.ReLU(),
                                   torch.nn.Conv2d(20, 8, 1, bias=True)).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(-1.0, 1.5); m.bias.data.normal_(0, 0.3)
    x0 = torch.randn(3, 16, 40, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    res = []
    default, default_bs = fused_bn.BN_IN_GEMM, fused_bn.BWD_STATS_IN_GEMM
    # (in-GEMM BatchNorm, gradient statistics in the data gradient's epilogue): the bit-identity is stated with the statistics
    # taken by the reduce operator on both sides; the epilogue form sums the same terms in another order (third run)
    for flag, bs in ((True, False), (False, False), (True, True)):
        fused_bn.BN_IN_GEMM, fused_bn.BWD_STATS_IN_GEMM = flag, bs
        try:
            m = copy.deepcopy(net)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
                loss = (y.float() * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum()
            loss.backward()
            res.append((y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                        {k: b.clone() for k, b in m.named_buffers()}))
        finally:
            fused_bn.BN_IN_GEMM, fused_bn.BWD_STATS_IN_GEMM = default, default_bs
    (ya, ga, pa, ba), (yb, gb, pb, bb), (yc, gc, pc, bc) = res
    assert ya.dtype == torch.bfloat16 and torch.equal(ya, yb) and torch.equal(ga, gb)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
    for k in ba:
        assert torch.equal(ba[k], bb[k]), k
    # statistics from the epilogue: the forward is untouched; gradients agree to the order of the fp32 partial sums (a changed last
    # bit of p / q may move a bf16 rounding of the input gradient)
    assert default_bs and torch.equal(ya, yc)
    for k in ba:
        assert torch.equal(ba[k], bc[k]), k
    assert float((ga.float() - gc.float()).abs().max()) <= 2.0 ** -7 * float(ga.float().abs().max())
    for k in pa:
        assert float((pa[k] - pc[k]).abs().max()) <= 1e-3 * float(pa[k].abs().max()) + 1e-6, k


@pytest.mark.parametrize("widths,x_grad,end_bn", [((16, 32, 20, 8), True, False), ((16, 32, 20, 8), False, False),
                                                  ((24, 160, 136, 96), True, False), ((72, 256, 256, 16), True, False),
                                                  ((16, 32, 20, 8), True, True), ((24, 160, 136, 96), True, True),
                                                  ((64, 128, 64, 256), True, True)])
@pytest.mark.gpu
def test_bn_relu_backward_inside_the_data_gradient_is_bit_identical(dev, widths, x_grad, end_bn):
    """Conv -> BN -> ReLU -> Conv -> BN -> ReLU -> Conv under bf16 autocast: with the BatchNorm + ReLU backward's elementwise half
    formed inside the data gradient of the layer before (pdm_tg_gemm_nt_dy, fused_bn.LAZY_BN_BACKWARD: the operator's pass
    over (dZ, Y) and one read of dY disappear) the input gradient and every parameter gradient are BIT-identical to the path
    that runs pdm_bn_relu_backward whole.  Narrow (128 x 64 tiles) and wide (128 x 128) outputs, padded widths, and a first
    layer whose input needs no gradient (the apply half then runs on its own).  end_bn: the stack ENDS in BatchNorm + ReLU —
    the stand-alone operator (_BnRelu) then leaves its elementwise half to the last contraction's data gradient the same way."""
    import copy
    from pdm_ssd_amd import fused_bn
    torch.manual_seed(11)
    c0, c1, c2, c3 = widths
    tail = [torch.nn.Conv2d(c2, c3, 1, bias=False), torch.nn.BatchNorm2d(c3), torch.nn.ReLU()] if end_bn else \
           [torch.nn.Conv2d(c2, c3, 1, bias=True)]
    net = fused_bn.TrainSequential(torch.nn.Conv2d(c0, c1, 1, bias=False), torch.nn.BatchNorm2d(c1), torch.nn.ReLU(),
                                   torch.nn.Conv2d(c1, c2, 1, bias=False), torch.nn.BatchNorm2d(c2), torch.nn.ReLU(),
                                   *tail).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(-1.0, 1.5); m.bias.data.normal_(0, 0.3)
    x0 = torch.randn(3, c0, 50, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    res = []
    default, default_bs = fused_bn.LAZY_BN_BACKWARD, fused_bn.BWD_STATS_IN_GEMM
    assert default and default_bs and fused_bn.BN_IN_GEMM, "the fused forms are the default path"
    # the bit-identity is stated with the gradient statistics taken by the reduce operator on both sides; with the statistics
    # taken in the data gradients' epilogues (the default, third run) the two forms tile their products differently and sum the
    # same terms in another order
    for flag, bs in ((True, False), (False, False), (True, True)):
        fused_bn.LAZY_BN_BACKWARD, fused_bn.BWD_STATS_IN_GEMM = flag, bs
        try:
            m = copy.deepcopy(net)
            x = x0.clone().requires_grad_(x_grad)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
                loss = (y.float() * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum()
            loss.backward()
            res.append((y.detach().clone(), x.grad.clone() if x_grad else None, {k: p.grad.clone() for k, p in m.named_parameters()}))
        finally:
            fused_bn.LAZY_BN_BACKWARD, fused_bn.BWD_STATS_IN_GEMM = default, default_bs
    (ya, ga, pa), (yb, gb, pb), (yc, gc, pc) = res
    assert torch.equal(ya, yb) and (not x_grad or torch.equal(ga, gb))
    for k in pa:
        assert torch.isfinite(pa[k]).all() and torch.equal(pa[k], pb[k]), k
    assert torch.equal(ya, yc)
    if x_grad:
        assert float((ga.float() - gc.float()).abs().max()) <= 2.0 ** -6 * float(ga.float().abs().max())
    for k in pa:
        assert torch.isfinite(pc[k]).all() and float((pa[k] - pc[k]).abs().max()) <= 2e-3 * float(pa[k].abs().max()) + 1e-6, k


@pytest.mark.gpu
def test_bn_relu_fp32_in_bf16_out_equals_the_fp32_operator_with_casts(dev):
    """The mixed form (fp32 input and input gradient, bf16 output and output gradient: pdm_bn_relu_forward / _backward with
    dtype 2) against the fp32 operator followed by `.to(bf16)` and fed a bf16 gradient cast to fp32: outputs, input gradient,
    dgamma, dbeta and running statistics BIT-identical."""
    from pdm_ssd_amd import fused_bn
    torch.manual_seed(3)
    B, C, H, W = 2, 64, 24, 20
    x0 = (torch.randn(B, C, H, W, device=dev) * 1.7 + 0.3).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    res = []
    for mixed in (True, False):
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        torch.manual_seed(4)                      # the same parameters for both forms
        with torch.no_grad():
            bn.weight.uniform_(-1.0, 1.5); bn.bias.normal_(0, 0.3)
        x = x0.clone().requires_grad_(True)
        y = fused_bn.batch_norm_relu(x, bn, True, None, out_bf16=mixed)
        if not mixed:
            assert y.dtype == torch.float32
            y = y.to(torch.bfloat16)
        assert y.dtype == torch.bfloat16
        y.backward(gy)
        res.append((y.detach().clone(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(),
                    bn.running_var.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("widths,x_grad", [((64, 64, 3), True), ((64, 64, 3), False), ((24, 136, 40), True), ((16, 32, 8), True)])
@pytest.mark.gpu
def test_relu_between_two_contractions_rides_in_the_kernels(dev, widths, x_grad):
    """Conv(bias) -> ReLU -> Conv(bias) under bf16 autocast (the heat-map head's output stack): with the ReLU applied in the second
    contraction's load path and its backward formed in the first one's data gradient (fused_bn._ReluRowsGemm: the BatchNorm + ReLU
    forms of the kernels with identity coefficients) outputs, input gradient and parameter gradients are BIT-identical to the path
    that runs torch's ReLU between the two contractions."""
    import copy
    from pdm_ssd_amd import fused_bn
    torch.manual_seed(21)
    c0, c1, c2 = widths
    net = fused_bn.TrainSequential(torch.nn.Conv2d(c0, c1, 1, bias=True), torch.nn.ReLU(), torch.nn.Conv2d(c1, c2, 1, bias=True)).to(dev).train()
    x0 = torch.randn(3, c0, 50, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    res = []
    default = fused_bn.BN_IN_GEMM
    for flag in (True, False):
        fused_bn.BN_IN_GEMM = flag
        try:
            m = copy.deepcopy(net)
            x = x0.clone().requires_grad_(x_grad)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
                loss = (y.float() * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum()
            loss.backward()
            res.append((y.detach().clone(), x.grad.clone() if x_grad else None, {k: p.grad.clone() for k, p in m.named_parameters()}))
        finally:
            fused_bn.BN_IN_GEMM = default
    (ya, ga, pa), (yb, gb, pb) = res
    assert torch.equal(ya, yb) and (not x_grad or torch.equal(ga, gb))
    for k in pa:
        assert torch.isfinite(pa[k]).all() and torch.equal(pa[k], pb[k]), k


@pytest.mark.parametrize("fused_opt", [True, False])
@pytest.mark.gpu
def test_training_over_several_optimizer_steps_is_the_same_with_and_without_the_weight_cache(dev, fused_opt):
    """Six AdamW steps of a Conv -> BN -> ReLU -> Conv -> BN -> ReLU -> Conv stack under bf16 autocast: with the packed bf16 weight
    pairs cached between steps (train_gemm.PACK_CACHE) every loss equals, BIT for bit, the run that repacks on every request.
    (torch's FUSED optimizers update parameters without moving their version counters: a cache that trusted the counter alone
    trained on the initial weights — losses then stop falling; this is the regression test of that.)"""
    import copy
    from pdm_ssd_amd import fused_bn, train_gemm as tg
    torch.manual_seed(31)
    net = fused_bn.TrainSequential(torch.nn.Conv2d(16, 32, 1, bias=False), torch.nn.BatchNorm2d(32), torch.nn.ReLU(),
                                   torch.nn.Conv2d(32, 64, 1, bias=False), torch.nn.BatchNorm2d(64), torch.nn.ReLU(),
                                   torch.nn.Conv2d(64, 8, 1, bias=True)).to(dev).train()
    x = torch.randn(4, 16, 40, 16, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    want = torch.randn(4, 8, 40, 16, device=dev)
    runs = []
    default = tg.PACK_CACHE
    for cache in (True, False):
        tg.PACK_CACHE = cache
        try:
            m = copy.deepcopy(net)
            opt = torch.optim.AdamW(m.parameters(), lr=3e-2, fused=fused_opt)
            losses = []
            for _ in range(6):
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = ((m(x).float() - want) ** 2).mean()
                loss.backward()
                opt.step()
                losses.append(float(loss))
            runs.append(losses)
        finally:
            tg.PACK_CACHE = default
    assert default and runs[0] == runs[1], runs
    assert runs[0][-1] < runs[0][0] and len(set(runs[0])) == 6, runs[0]     # and it does train (every step sees new weights)


@pytest.mark.parametrize("widths,ns", [((8, 16, 32), 16), ((16, 32, 64), 32), ((72, 128, 128), 16), ((24, 200, 256), 32)])
@pytest.mark.gpu
def test_pooled_tail_with_its_statistics_from_the_last_contraction(dev, widths, ns):
    """forward_max_pooled of an SA-like stack under bf16 autocast: with the pooled operator's statistics (column sums, group extremes
    and their indices) taken in the last contraction's epilogue (fused_bn.POOL_IN_GEMM, pdm_tg_gemm_nt_pool) against the operator's own
    pass over the tensor: the same pooled elements are selected; outputs and gradients agree to the order of the fp32 sums behind
    mean / variance (plain sums of the epilogue against pivoted sums of the pass: one bf16 rounding of the output)."""
    import copy
    from pdm_ssd_amd import fused_bn
    torch.manual_seed(17)
    c0, c1, c2 = widths
    net = fused_bn.TrainSequential(torch.nn.Conv2d(c0, c1, 1, bias=False), torch.nn.BatchNorm2d(c1), torch.nn.ReLU(),
                                   torch.nn.Conv2d(c1, c2, 1, bias=False), torch.nn.BatchNorm2d(c2), torch.nn.ReLU()).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(-1.0, 1.5); m.bias.data.normal_(0, 0.3)
    x0 = torch.randn(3, c0, 40, ns, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    res = []
    default = fused_bn.POOL_IN_GEMM          # (opt-in: measured no faster, fused_bn.py)
    for flag in (True, False):
        fused_bn.POOL_IN_GEMM = flag
        try:
            m = copy.deepcopy(net)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m.forward_max_pooled(x)
                loss = (y.float() * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum()
            loss.backward()
            res.append((y.detach().float().clone(), x.grad.float().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                        {k: b.clone() for k, b in m.named_buffers()}))
        finally:
            fused_bn.POOL_IN_GEMM = default
    (ya, ga, pa, ba), (yb, gb, pb, bb) = res
    assert ya.shape == (3, c2, 40, 1)
    assert float((ya - yb).abs().max()) <= 2.0 ** -7 * float(yb.abs().max())
    assert float((ga - gb).abs().max()) <= 2.0 ** -5 * float(gb.abs().max())
    for k in pa:
        assert float((pa[k] - pb[k]).abs().max()) <= 5e-3 * float(pb[k].abs().max()) + 1e-6, k
    for k in ba:
        assert torch.allclose(ba[k].float(), bb[k].float(), rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("widths", [(128, 64, 8), (32, 40, 16)])
@pytest.mark.gpu
def test_bn_relu_on_a_tensor_that_no_contraction_produced_rides_in_the_next_contraction(dev, widths):
    """BN -> ReLU -> Conv -> BN -> ReLU -> Conv on a bf16 channels-last tensor handed in from outside (the heat-map head behind its
    depthwise convolution): with fused_bn.BN_FROM_X the first BatchNorm's statistics pass runs alone (pdm_bn_forward_coef) and its apply
    half rides in the contraction's load path — outputs, running statistics and (statistics by the reduce operator on both sides) every
    gradient BIT-identical to the stand-alone operator; with the gradient statistics from the epilogue (the default) equal to summation
    order."""
    import copy
    from pdm_ssd_amd import fused_bn
    torch.manual_seed(23)
    c0, c1, c2 = widths
    net = fused_bn.TrainSequential(torch.nn.BatchNorm2d(c0), torch.nn.ReLU(), torch.nn.Conv2d(c0, c1, 1, bias=False),
                                   torch.nn.BatchNorm2d(c1), torch.nn.ReLU(), torch.nn.Conv2d(c1, c2, 1, bias=True)).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(-1.0, 1.5); m.bias.data.normal_(0, 0.3)
    x0 = (torch.randn(3, c0, 40, 16, device=dev) * 1.3 + 0.2).bfloat16().contiguous(memory_format=torch.channels_last)
    res = []
    d_x, d_bs = fused_bn.BN_FROM_X, fused_bn.BWD_STATS_IN_GEMM
    assert d_x and d_bs
    for from_x, bs in ((True, False), (False, False), (True, True)):
        fused_bn.BN_FROM_X, fused_bn.BWD_STATS_IN_GEMM = from_x, bs
        try:
            m = copy.deepcopy(net)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
                loss = (y.float() * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum()
            loss.backward()
            res.append((y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                        {k: b.clone() for k, b in m.named_buffers()}))
        finally:
            fused_bn.BN_FROM_X, fused_bn.BWD_STATS_IN_GEMM = d_x, d_bs
    (ya, ga, pa, ba), (yb, gb, pb, bb), (yc, gc, pc, bc) = res
    assert torch.equal(ya, yb) and torch.equal(ga, gb) and torch.equal(ya, yc)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
        assert float((pa[k] - pc[k]).abs().max()) <= 2e-3 * float(pa[k].abs().max()) + 1e-6, k
    for k in ba:
        assert torch.equal(ba[k], bb[k]) and torch.equal(ba[k], bc[k]), k
    assert float((ga.float() - gc.float()).abs().max()) <= 2.0 ** -6 * float(ga.float().abs().max())
