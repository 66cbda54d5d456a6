"""Training-mode BatchNorm + ReLU as ONE operator (HIP kernels of csrc/bn_relu.hip) and the Sequential that uses it.

The reference builds its shared MLPs as torch triples Conv/Linear -> BatchNorm -> ReLU
(/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:19-55,
models/dense_heads/point_head_template.py:35-48); in a train step on MI355X the BatchNorm and ReLU kernels of
torch / MIOpen took about a quarter of the device time at a fraction of the HBM rate (profiles/).  `TrainSequential` is
a drop-in `nn.Sequential` (same child indices, hence the same state_dict keys) that, in training mode on the GPU,
runs every (BatchNorm, ReLU) pair after a convolution / linear layer as one forward kernel pair and one backward
kernel pair, statistics in fp32, activations fp32 or bf16 (autocast).  Anything it does not recognise — eval mode,
CPU tensors, exotic shapes, `momentum=None`, `track_running_stats=False` — goes through the torch modules unchanged.
"""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _native

ENABLED = os.environ.get("PDM_FUSED_BN", "1") != "0"


def _layout(x):
    """(layout, n, L) of the C-ABI for a dense (N, C, ...) tensor, or None when neither layout applies."""
    if x.dim() < 2 or x.numel() == 0:
        return None
    C = x.shape[1]
    v = 8 if x.dtype == torch.bfloat16 else 4
    if x.dim() == 2:
        return (0, x.shape[0], 1) if x.is_contiguous() and C % v == 0 and C // v <= 256 else None
    L = x.numel() // (x.shape[0] * C)
    if x.is_contiguous():
        if L == 1:
            return (0, x.shape[0], 1) if C % v == 0 and C // v <= 256 else None
        return (1, x.shape[0], L) if L % v == 0 and C <= 65535 else None
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return (0, x.numel() // C, 1) if C % v == 0 and C // v <= 256 else None
    return None


class _BnRelu(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, relu, layout, n, L):
        C = x.shape[1]
        dtype = 1 if x.dtype == torch.bfloat16 else 0
        y = torch.empty_like(x)
        coef = torch.empty((4, C), dtype=torch.float32, device=x.device)
        parts = _native.lib().pdm_bn_parts(layout, n, C, L)
        partial = torch.empty((parts, C, 2), dtype=torch.float32, device=x.device)
        _native.call("pdm_bn_relu_forward", torch.cuda.current_stream(x.device).cuda_stream, dtype, layout, n, C, L, x.data_ptr(),
                     y.data_ptr(), weight.data_ptr(), bias.data_ptr(), float(eps), float(momentum),
                     0 if running_mean is None else running_mean.data_ptr(), 0 if running_var is None else running_var.data_ptr(),
                     coef.data_ptr(), partial.data_ptr(), int(relu))
        ctx.save_for_backward(x, coef)
        ctx.meta = (dtype, layout, n, C, L, int(relu), parts)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, coef = ctx.saved_tensors
        dtype, layout, n, C, L, relu, parts = ctx.meta
        if dy.dtype != x.dtype or dy.stride() != x.stride():
            dy = torch.empty_like(x).copy_(dy)       # same type and memory format as x
        dx = torch.empty_like(x)
        grads = torch.empty((4, C), dtype=torch.float32, device=x.device)
        partial = torch.empty((parts, C, 2), dtype=torch.float32, device=x.device)
        _native.call("pdm_bn_relu_backward", torch.cuda.current_stream(x.device).cuda_stream, dtype, layout, n, C, L, x.data_ptr(),
                     dy.data_ptr(), dx.data_ptr(), coef.data_ptr(), grads.data_ptr(), partial.data_ptr(), relu)
        return dx, grads[0], grads[1], None, None, None, None, None, None, None, None


class _BnReluPool(Function):
    """relu(bn(x)) max-pooled over the last axis of a channels-last (B, C, M, ns) tensor -> (B, C, M, 1), as one operator
    (csrc/bn_relu.hip, "BatchNorm + ReLU + max over the ns neighbours").  Ties go to the first neighbour attaining the
    extreme of x (torch's max_pool2d picks the first maximum of the ROUNDED output: the same element unless two
    different inputs round to one bf16 output)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum):
        B, C, M, ns = x.shape
        G = B * M
        dtype = 1 if x.dtype == torch.bfloat16 else 0
        dev = x.device
        y = torch.empty((B, M, 1, C), dtype=x.dtype, device=dev)
        keep = torch.empty((2, G, C), dtype=x.dtype, device=dev)
        idx = torch.empty((2, G, C), dtype=torch.uint8, device=dev)
        coef = torch.empty((4, C), dtype=torch.float32, device=dev)
        parts = _native.lib().pdm_bn_pool_parts(dtype, G, C)
        partial = torch.empty((parts, C, 2), dtype=torch.float32, device=dev)
        _native.call("pdm_bn_relu_pool_forward", torch.cuda.current_stream(dev).cuda_stream, dtype, G, ns, C, x.data_ptr(), y.data_ptr(),
                     keep[0].data_ptr(), keep[1].data_ptr(), idx[0].data_ptr(), idx[1].data_ptr(), weight.data_ptr(), bias.data_ptr(),
                     float(eps), float(momentum), running_mean.data_ptr(), running_var.data_ptr(), coef.data_ptr(),
                     partial.data_ptr(), 1)
        ctx.save_for_backward(x, keep, idx, coef)
        ctx.meta = (dtype, G, ns, C, parts)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, keep, idx, coef = ctx.saved_tensors
        dtype, G, ns, C, parts = ctx.meta
        dyp = dy.permute(0, 2, 3, 1).to(x.dtype).contiguous()       # (B, M, 1, C): rows x channels
        dx = torch.empty_like(x)
        grads = torch.empty((4, C), dtype=torch.float32, device=x.device)
        partial = torch.empty((parts, C, 2), dtype=torch.float32, device=x.device)
        _native.call("pdm_bn_relu_pool_backward", torch.cuda.current_stream(x.device).cuda_stream, dtype, G, ns, C, x.data_ptr(),
                     dyp.data_ptr(), dx.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), idx[0].data_ptr(), idx[1].data_ptr(),
                     coef.data_ptr(), grads.data_ptr(), partial.data_ptr(), 1)
        return dx, grads[0], grads[1], None, None, None, None


def pool_applies(x, bn):
    """the pooled operator takes channels-last (B, C, M, ns) tensors, ns <= 255"""
    v = 8 if x.dtype == torch.bfloat16 else 4
    return (applies(x, bn) and x.dim() == 4 and x.shape[3] <= 255 and x.is_contiguous(memory_format=torch.channels_last)
            and not x.is_contiguous() and x.shape[1] % v == 0 and x.shape[1] // v <= 256)


def applies(x, bn):
    return (ENABLED and bn.training and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and bn.affine
            and bn.track_running_stats and bn.momentum is not None and bn.weight.dtype == torch.float32
            and x.dim() >= 2 and x.shape[1] == bn.num_features and _layout(x) is not None)


def batch_norm_relu(x, bn, relu=True):
    """bn(x) followed by ReLU (relu=True), through the fused kernels when `applies`, else through torch."""
    if not applies(x, bn):
        y = bn(x)
        return torch.relu(y) if relu else y
    layout, n, L = _layout(x)
    with torch.no_grad():
        bn.num_batches_tracked += 1
    return _BnRelu.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu, layout, n, L)


class _LinearSplitK(Function):
    """y = x W^T (+ b) for VERY tall x (the point head: 524288 rows x 128-256 channels).  The weight gradient
    dW = dy^T x contracts over the rows; the vendor GEMM gives that shape (256 x 256 outputs, K = 524288) a handful of
    workgroups (measured 0.86-1.5 ms per layer).  Here the rows are cut into slabs, one batched GEMM forms a partial dW per
    slab, and the partials are summed in fp32."""
    SLAB = 8192

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        # operands rounded to bf16 here (what autocast would do), so the fp32 parameters get fp32 gradients straight
        # from the fp32 slab sum: no bf16 round trip of the gradient, three small cast kernels fewer per layer
        xb, wb = x.to(torch.bfloat16), weight.to(torch.bfloat16)
        ctx.save_for_backward(xb, wb)
        ctx.has_bias, ctx.xdtype = bias is not None, x.dtype
        return torch.nn.functional.linear(xb, wb, None if bias is None else bias.to(torch.bfloat16))

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = (dy @ weight).to(ctx.xdtype) if ctx.needs_input_grad[0] else None
        s = x.shape[0] // _LinearSplitK.SLAB
        dw = torch.bmm(dy.view(s, _LinearSplitK.SLAB, -1).transpose(1, 2), x.view(s, _LinearSplitK.SLAB, -1)).sum(0, dtype=torch.float32)
        db = dy.sum(0, dtype=torch.float32) if ctx.has_bias else None
        return dx, dw, db


def tall_linear(x, lin):
    """lin(x) with the split-K weight gradient when x is a tall bf16-autocast matrix on the GPU."""
    if (ENABLED and x.is_cuda and x.dim() == 2 and x.is_contiguous() and x.shape[0] >= 8 * _LinearSplitK.SLAB
            and x.shape[0] % _LinearSplitK.SLAB == 0 and _bf16_autocast() and lin.weight.requires_grad
            and lin.weight.dtype == torch.float32):
        return _LinearSplitK.apply(x, lin.weight, lin.bias)
    return lin(x)


def _bf16_autocast():
    return torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16


def _slab(rows):
    """largest power-of-two slab <= 16384 that divides `rows` into at least 8 slabs, else 0"""
    s = 16384
    while s >= 512:
        if rows % s == 0 and rows // s >= 8:
            return s
        s //= 2
    return 0


class _Conv1x1SplitK(Function):
    """1x1 convolution (the shared MLPs' layers) as three GEMMs over the tensor's own storage: a channels-last (B, C, H, W)
    tensor IS a (positions, C) matrix and a position-fastest (B, C, L) tensor is B (C, L) matrices, so the forward and
    the input gradient are plain (batched) GEMMs and the WEIGHT gradient (dW = sum over positions of dy x^T) is a batched
    GEMM over slabs of positions summed in fp32.  MIOpen's weight-gradient kernels with their fp32 workspace fills, casts
    and transposes, and the zero-fills in front of its data-gradient kernels, took ~6 ms of the train step."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight):
        xb, wb = x.to(torch.bfloat16), weight.to(torch.bfloat16)   # autocast's rounding, done here (see _LinearSplitK)
        ctx.save_for_backward(xb, wb)
        ctx.xdtype = x.dtype
        cout, cin = weight.shape[0], weight.shape[1]
        w2 = wb.view(cout, cin)
        ctx.cl = xb.dim() == 4 and xb.is_contiguous(memory_format=torch.channels_last) and not xb.is_contiguous()
        if ctx.cl:
            B, _, H, W = xb.shape
            return (xb.permute(0, 2, 3, 1).reshape(-1, cin) @ w2.t()).view(B, H, W, cout).permute(0, 3, 1, 2)
        xr = xb.contiguous().reshape(xb.shape[0], cin, -1)
        return torch.matmul(w2, xr).view(xb.shape[0], cout, *xb.shape[2:])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        w2 = weight.view(cout, cin)
        dy = dy.to(torch.bfloat16)
        dx = None
        if ctx.cl:
            B, _, H, W = x.shape
            xv = x.permute(0, 2, 3, 1).reshape(-1, cin)                       # views: positions x channels
            gv = dy.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1, cout)
            if ctx.needs_input_grad[0]:
                dx = (gv @ w2).view(B, H, W, cin).permute(0, 3, 1, 2).to(ctx.xdtype)
            s = _slab(xv.shape[0])
            if s:
                dw = torch.bmm(gv.view(-1, s, cout).transpose(1, 2), xv.view(-1, s, cin)).sum(0, dtype=torch.float32)
            else:
                dw = (gv.t() @ xv).float()
        else:                                                                   # (B, C, L...) position fastest: one GEMM per sample
            xr = x.contiguous().reshape(x.shape[0], cin, -1)
            gb = dy.contiguous().reshape(x.shape[0], cout, -1)
            if ctx.needs_input_grad[0]:
                dx = torch.matmul(w2.t(), gb).view(x.shape).to(ctx.xdtype)
            dw = torch.bmm(gb, xr.transpose(1, 2)).sum(0, dtype=torch.float32)
        return dx, dw.view(cout, cin, *([1] * (x.dim() - 2)))


def conv1x1(x, conv):
    """conv(x) for a plain 1x1 convolution, with the split-K weight gradient under bf16 autocast on the GPU."""
    if (ENABLED and x.is_cuda and _bf16_autocast() and conv.bias is None and conv.weight.requires_grad
            and conv.weight.dtype == torch.float32
            and all(k == 1 for k in conv.kernel_size) and all(v == 1 for v in conv.stride) and all(v == 0 for v in conv.padding)
            and all(v == 1 for v in conv.dilation) and conv.groups == 1 and isinstance(conv.padding, tuple)
            and x.dim() == conv.weight.dim() and x.numel() >= (1 << 20)):
        return _Conv1x1SplitK.apply(x, conv.weight)
    return conv(x)


_BN = (nn.BatchNorm1d, nn.BatchNorm2d)


class TrainSequential(nn.Sequential):
    """nn.Sequential whose (BatchNorm, ReLU) pairs run fused in training mode on the GPU (see the module docstring)."""

    def forward_max_pooled(self, x):
        """F.max_pool2d(self(x), kernel_size=[1, x.size(3)]) for a stack that ends in (BatchNorm2d, ReLU): in training
        mode on the GPU the last pair and the pooling run as one operator (the normalised tensor is never written)."""
        mods = list(self)
        if (ENABLED and self.training and x.is_cuda and len(mods) >= 3 and isinstance(mods[-1], nn.ReLU)
                and isinstance(mods[-2], nn.BatchNorm2d)):
            h = self._run(x, mods[:-2])
            bn = mods[-2]
            if pool_applies(h, bn):
                with torch.no_grad():
                    bn.num_batches_tracked += 1
                return _BnReluPool.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)
            h = self._run(h, mods[-2:])
        else:
            h = self(x)
        return torch.nn.functional.max_pool2d(h, kernel_size=[1, h.size(3)])

    def forward(self, x):
        if not (ENABLED and self.training and x.is_cuda):
            return super().forward(x)
        return self._run(x, list(self))

    @staticmethod
    def _run(x, mods):
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _BN) and applies(x, m):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = batch_norm_relu(x, m, relu)
                i += 2 if relu else 1
            elif isinstance(m, nn.Linear):
                x = tall_linear(x, m)
                i += 1
            elif type(m) in (nn.Conv1d, nn.Conv2d):
                x = conv1x1(x, m)
                i += 1
            else:
                x = m(x)
                i += 1
        return x
