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
    if x.dim() == 3 and not x.is_contiguous() and x.transpose(1, 2).is_contiguous():
        # (B, C, L) view of point-major (B, L, C) storage (the rows layout of the training path): rows x C
        return (0, x.shape[0] * x.shape[2], 1) if C % v == 0 and C // v <= 256 else None
    if x.is_contiguous():
        if L == 1:
            return (0, x.shape[0], 1) if C % v == 0 and C // v <= 256 else None
        return (1, x.shape[0], L) if L % v == 0 and C <= 65535 else None
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return (0, x.numel() // C, 1) if C % v == 0 and C // v <= 256 else None
    return None


def _same_strides(a, b):
    """Same memory layout: equal strides on every dimension longer than one (the stride of a length-1 dimension is free — the
    (B, C, N, 1) gradients of the FP stacks arrive with 1 where the activation has C, which cost a 243 MB copy per step)."""
    return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)


class _BnRelu(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, relu, layout, n, L, stats=None, out_bf16=False, link=None):
        ctx.link = link      # shared with the rows GEMM node that produced x (TrainSequential._run): see backward
        C = x.shape[1]
        dtype = 1 if x.dtype == torch.bfloat16 else 0
        if out_bf16 and dtype == 0 and layout == 0:
            # fp32 input, bf16 output (and bf16 gradient in, fp32 gradient out): what `bn(x).relu().to(bf16)` and its backward
            # compute, bit for bit, without the two cast passes — for an fp32 map that a contraction under autocast reads next
            dtype = 2
        y = torch.empty_like(x, dtype=torch.bfloat16) if dtype == 2 else torch.empty_like(x)
        coef = torch.empty((4, C), dtype=torch.float32, device=x.device)
        parts = _native.lib().pdm_bn_parts(layout, n, C, L)
        if stats is not None and layout == 0:
            # the producing GEMM left the column sums of x and x^2 (train_gemm.gemm_nt(..., stats=True)): no statistics pass
            assert stats.shape[1:] == (C, 2) and stats.dtype == torch.float32 and stats.is_contiguous()
            _native.call("pdm_bn_relu_forward_stats", torch.cuda.current_stream(x.device).cuda_stream, dtype, n, C, x.data_ptr(),
                         y.data_ptr(), weight.data_ptr(), bias.data_ptr(), float(eps), float(momentum),
                         0 if running_mean is None else running_mean.data_ptr(), 0 if running_var is None else running_var.data_ptr(),
                         coef.data_ptr(), stats.data_ptr(), stats.shape[0], int(relu))
        else:
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
        want = torch.bfloat16 if dtype == 2 else x.dtype
        if dy.dtype != want or not _same_strides(dy, x):
            dy = torch.empty_like(x, dtype=want).copy_(dy)       # the operator's gradient type, x's memory format
        grads = torch.empty((4, C), dtype=torch.float32, device=x.device)
        partial = torch.empty((parts, C, 2), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if ctx.link is not None and LAZY_BN_BACKWARD and dtype == 1 and layout == 0 and relu:
            # x came straight from a rows GEMM node: only the gradient statistics are taken here; that node's data gradient
            # forms dx while it reads dy and x (see _take_lazy_bn_backward — the same hand-over _BnReluRowsGemm uses)
            from . import train_gemm as tg
            xr, dyr = tg.row_view(x), tg.row_view(dy)
            if xr is not None and dyr is not None and xr.shape == dyr.shape and xr.stride(0) == C and dyr.stride(0) == C:
                _native.call("pdm_bn_relu_backward_stats", stream, 1, 0, n, C, 1, x.data_ptr(), dy.data_ptr(), coef.data_ptr(),
                             grads.data_ptr(), partial.data_ptr(), 1)
                ctx.link['lazy'] = (xr, coef, grads, dyr.data_ptr())
                return dy, grads[0], grads[1], None, None, None, None, None, None, None, None, None, None, None
        dx = torch.empty_like(x)
        _native.call("pdm_bn_relu_backward", stream, dtype, layout, n, C, L, x.data_ptr(),
                     dy.data_ptr(), dx.data_ptr(), coef.data_ptr(), grads.data_ptr(), partial.data_ptr(), relu)
        return dx, grads[0], grads[1], None, None, None, None, None, None, None, None, None, None, None


class _BnReluPool(Function):
    """relu(bn(x)) max-pooled over the last axis of a channels-last (B, C, M, ns) tensor -> (B, C, M, 1), as one operator
    (csrc/bn_relu.hip, "BatchNorm + ReLU + max over the ns neighbours").  Ties go to the first neighbour attaining the
    extreme of x (torch's max_pool2d picks the first maximum of the ROUNDED output: the same element unless two
    different inputs round to one bf16 output)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, stats=None, keep=None, idx=None):
        B, C, M, ns = x.shape
        G = B * M
        dtype = 1 if x.dtype == torch.bfloat16 else 0
        dev = x.device
        y = torch.empty((B, M, 1, C), dtype=x.dtype, device=dev)
        coef = torch.empty((4, C), dtype=torch.float32, device=dev)
        parts = _native.lib().pdm_bn_pool_parts(dtype, G, C)
        if stats is not None:
            # the contraction that produced x left the column sums AND every group's extremes in its epilogue (pdm_tg_gemm_nt_pool):
            # no statistics pass over x (POOL_IN_GEMM)
            assert dtype == 1 and keep.shape == (2, G, C) and idx.shape == (2, G, C) and stats.shape[1:] == (C, 2)
            _native.call("pdm_bn_relu_pool_forward_kept", torch.cuda.current_stream(dev).cuda_stream, 1, G, ns, C, y.data_ptr(), keep[0].data_ptr(),
                         keep[1].data_ptr(), weight.data_ptr(), bias.data_ptr(), float(eps), float(momentum), running_mean.data_ptr(),
                         running_var.data_ptr(), coef.data_ptr(), stats.data_ptr(), stats.shape[0], 1)
        else:
            keep = torch.empty((2, G, C), dtype=x.dtype, device=dev)
            idx = torch.empty((2, G, C), dtype=torch.uint8, device=dev)
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
        return dx, grads[0], grads[1], None, None, None, None, None, None, None


# 1: the last contraction of an SA scale leaves the pooled operator's statistics (column sums, group extremes + indices) in its
# epilogue (pdm_tg_gemm_nt_pool): the operator's own pass over the (B M ns, C) tensor disappears.  Built, exact, measured and NOT
# faster: 18.77-18.81 ms per step against 18.53-18.62 (A/B twice on one box) — the epilogue's compare / select work per element does
# not overlap the tile's memory traffic at two workgroups per CU, and costs the eight widest-row contractions more than the 0.4 ms
# pass it removes (with one thread per (group, chunk) instead of a quad: 18.88).  Off by default.
POOL_IN_GEMM = os.environ.get("PDM_POOL_IN_GEMM", "0") == "1"


def pool_applies(x, bn):
    """the pooled operator takes channels-last (B, C, M, ns) tensors, ns <= 255"""
    v = 8 if x.dtype == torch.bfloat16 else 4
    return (applies(x, bn) and x.dim() == 4 and x.shape[3] <= 255 and x.is_contiguous(memory_format=torch.channels_last)
            and not x.is_contiguous() and x.shape[1] % v == 0 and x.shape[1] // v <= 256)


def applies(x, bn):
    return (ENABLED and bn.training and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and bn.affine
            and bn.track_running_stats and bn.momentum is not None and bn.weight.dtype == torch.float32
            and x.dim() >= 2 and x.shape[1] == bn.num_features and _layout(x) is not None)


def _padded_applies(x, bn):
    """x carries bn.num_features channels rounded up to a multiple of 8, the extra ones ZERO (the rows layers keep 16-byte
    rows that way, _RowsGemm): BatchNorm runs over the padded width with zero gamma / beta on the padding, which then
    stays zero forward and backward."""
    C = bn.num_features
    return (ENABLED and bn.training and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and bn.affine
            and bn.track_running_stats and bn.momentum is not None and bn.weight.dtype == torch.float32
            and x.dim() >= 2 and C % 8 != 0 and x.shape[1] == _round8(C) and _layout(x) is not None)


# BatchNorm's num_batches_tracked += 1 is one tiny launch per layer and step; inside TrainSequential._run the layers of a stack
# are collected here and bumped with ONE multi-tensor add at the end of the stack (same values, ~26 launches fewer per step)
_bump_later = []


def _bump(bn):
    if _bump_later and _bump_later[-1] is not None:
        _bump_later[-1].append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


class counter_scope:
    """`with fused_bn.counter_scope():` around a model's forward: every BatchNorm counter inside (all stacks, the pooled operators)
    is bumped by ONE multi-tensor add when the scope closes instead of one per stack (the detector: 24 launches -> 1)."""

    def __enter__(self):
        _bump_later.append([])
        return self

    def __exit__(self, *exc):
        counters = _bump_later.pop()
        if counters:
            with torch.no_grad():
                torch._foreach_add_(counters, 1)
        return False


def batch_norm_relu(x, bn, relu=True, stats=None, out_bf16=False, link=None):
    """bn(x) followed by ReLU (relu=True), through the fused kernels when `applies`, else through torch.
    stats: the column sums the producing GEMM took of x ([tiles][C][2], rows_linear(..., want_stats=True)) or None."""
    if _padded_applies(x, bn):
        C, Cp = bn.num_features, x.shape[1]
        z = bn.weight.new_zeros(Cp - C)
        rm, rv = torch.cat([bn.running_mean, z]), torch.cat([bn.running_var, z + 1.0])
        layout, n, L = _layout(x)
        y = _BnRelu.apply(x, torch.cat([bn.weight, z]), torch.cat([bn.bias, z]), rm, rv, bn.eps, bn.momentum, relu, layout, n, L,
                          stats if layout == 0 else None, False, link if layout == 0 else None)
        with torch.no_grad():
            bn.running_mean.copy_(rm[:C]); bn.running_var.copy_(rv[:C])
            _bump(bn)
        return y
    if not applies(x, bn):
        y = bn(x)
        return torch.relu(y) if relu else y
    layout, n, L = _layout(x)
    with torch.no_grad():
        _bump(bn)
    return _BnRelu.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu, layout, n, L,
                         stats if layout == 0 else None, bool(out_bf16), link if layout == 0 else None)


def _round8(v):
    return (v + 7) // 8 * 8


# 1 (default): between two contractions of a stack the BatchNorm + ReLU backward's elementwise half rides in the data gradient of
# the layer before (pdm_tg_gemm_nt_dy): no pass over (dZ, Y) of its own and one read of dY less.  Bit-identical gradients.
LAZY_BN_BACKWARD = os.environ.get("PDM_LAZY_BN_BACKWARD", "1") == "1"


# 1 (default): the statistics of a BatchNorm + ReLU backward between two contractions of a stack (sum g, sum g xhat over the gradient
# the next layer's data gradient produces) are taken in THAT contraction's epilogue (pdm_tg_gemm_nt_bs / pdm_tg_gemm_nt_dy_bs): the
# reduce pass over (gradient, x) — a second read of both — disappears; only the finalize launch is left.
BWD_STATS_IN_GEMM = os.environ.get("PDM_BWD_STATS_IN_GEMM", "1") == "1"


def _take_lazy_bn_backward(link, dyr, wt, want_dx, bs=None):
    """The gradient rows `dyr` of a contraction's output may be UNFORMED: when the BatchNorm + ReLU behind it belongs to a
    _BnReluRowsGemm, that node hands back the gradient of relu(bn(y)) as it is and leaves (y, coef, grads) in the link it shares
    with the producer of y — this node.  Returns (dy rows formed, dx rows or None): with want_dx the data gradient forms dy on
    the way (pdm_tg_gemm_nt_dy), otherwise the BatchNorm operator's apply half does.
    bs = (x rows, coef) of the BatchNorm + ReLU in FRONT of this layer (its output gradient is this node's dx): when the data
    gradient is formed here, its epilogue takes that BatchNorm's gradient statistics; returned as a third value (or None)."""
    lazy = link.pop('lazy', None) if link is not None else None
    if lazy is None:
        return dyr, None, None
    from . import train_gemm as tg
    y_rows, coef, grads, ptr = lazy
    if dyr.data_ptr() != ptr or dyr.shape != y_rows.shape or dyr.dtype != torch.bfloat16:
        raise RuntimeError("fused_bn: an unformed BatchNorm gradient did not reach the contraction it was left for "
                           f"(rows {tuple(dyr.shape)} at {dyr.data_ptr():#x}, expected {tuple(y_rows.shape)} at {ptr:#x})")
    R, K = y_rows.shape
    # per shape (tools/diag/lazy_bn_rate.py): the fused form wins 1.1-1.5x where the data gradient has at most two column tiles
    # (N <= 256) and rows of >= 64 bytes; 16-channel rows (0.69x) and three or more column tiles (every tile re-reads and
    # re-forms the operand: 0.72-0.81x) take the apply half of the operator and the plain contraction
    if want_dx and 32 <= K <= 512 and wt.shape[0] <= 256:
        if bs is not None:
            dxr, dy_formed, partial = tg.gemm_nt_dy(dyr, y_rows, coef, grads, wt, bs=bs)
            return dy_formed, dxr, partial
        dxr, dy_formed = tg.gemm_nt_dy(dyr, y_rows, coef, grads, wt)
        return dy_formed, dxr, None
    dy_formed = torch.empty_like(y_rows)
    _native.call("pdm_bn_relu_backward_apply", torch.cuda.current_stream(dyr.device).cuda_stream, 1, 0, R, K, 1, y_rows.data_ptr(),
                 dyr.data_ptr(), dy_formed.data_ptr(), coef.data_ptr(), grads.data_ptr(), 1)
    return dy_formed, None, None


class _RowsGemm(Function):
    """1x1 convolution / Linear over channels-last rows on this library's bf16 MFMA kernels (csrc/train_gemm.hip):
    forward y = x W^T [+ b], data gradient dx = dy W, weight gradient dW = dy^T x — no vendor GEMM, no layout or dtype
    copy of an activation that is already bf16 rows.  Numerics: operands rounded to bf16 (what autocast does), fp32
    accumulation, y and dx rounded once to bf16, dW / db fp32 straight into the fp32 parameters.

    x: (R, K) rows, a channels-last (B, K, H, W) tensor or a (B, K, L) tensor stored (B, L, K).  K may exceed the layer's
    in_channels up to the next multiple of 8: those channels are ZERO PADDING by contract (the grouping operator writes
    them) and meet zero weight columns.  Returns (y, stats): y in x's layout with N channels (a view of (R, round8(N))
    storage), stats = the per-row-tile column sums of y and y^2 for the BatchNorm that follows (or None)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, want_stats, keep_pad=False, link=None, pool_ns=0):
        from . import train_gemm as tg
        ctx.link = link
        ctx.set_materialize_grads(False)     # no zero tensor (a fill launch per node and step) for the statistics output's gradient
        xr = tg.row_view(x)
        assert xr is not None
        if xr.dtype != torch.bfloat16:
            xr = xr.to(torch.bfloat16)                       # autocast's rounding of an fp32 activation
        R, K = xr.shape
        N = weight.shape[0]
        w2 = weight.reshape(N, -1)
        Kw, Np = w2.shape[1], _round8(N)
        assert Kw <= K < Kw + 8 and K % 8 == 0
        wb, wt = tg.pack_weight_pair(w2, Np, K)              # forward weights and their transpose (data gradient): one launch
        if bias is not None and Np != N:
            bias = torch.cat([bias.detach().float(), bias.new_zeros(Np - N, dtype=torch.float32)])
        if pool_ns and want_stats and link is not None:
            y, stats, link['pool'] = tg.gemm_nt(xr, wb, bias=bias, stats=True, pool_ns=pool_ns)     # (keeps travel beside the autograd graph)
        else:
            y, stats = tg.gemm_nt(xr, wb, bias=bias, stats=True) if want_stats else (tg.gemm_nt(xr, wb, bias=bias), None)
        ctx.save_for_backward(xr, weight, wt)
        ctx.geom = (tuple(x.shape), x.dim(), N, Np, K, Kw, bias is not None, x.dtype)
        # keep_pad: the caller (a BatchNorm over the padded width follows) takes all round8(N) channels, the extra ones zero
        out = _rows_to_layout(y, x, Np if keep_pad else N)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy, _dstats=None):
        from . import train_gemm as tg
        if dy is None:
            return None, None, None, None, None, None, None
        xr, weight, wt = ctx.saved_tensors
        xshape, xdim, N, Np, K, Kw, has_bias, xdtype = ctx.geom
        R = xr.shape[0]
        dyr = tg.row_view(dy)
        if dyr is None or dyr.dtype != torch.bfloat16 or dyr.shape[1] != Np or dyr.stride(0) % 8:
            # a gradient that does not arrive as bf16 rows of the padded width: one copy into that form
            nc = dy.shape[1]                                  # N, or Np when the forward kept the padding
            src = dy.movedim(1, -1).reshape(R, nc) if xdim > 2 else dy
            dyr = torch.zeros((R, Np), dtype=torch.bfloat16, device=dy.device) if Np != nc else torch.empty((R, Np), dtype=torch.bfloat16, device=dy.device)
            dyr[:, :nc].copy_(src)
        dx = None
        dyr, dxr, _ = _take_lazy_bn_backward(ctx.link, dyr, wt, ctx.needs_input_grad[0])
        if ctx.needs_input_grad[0]:
            if dxr is None:
                dxr = tg.gemm_nt(dyr, wt)                    # (R, K) bf16: the pad channels come out zero
            dx = _rows_to_layout(dxr, None, K, xshape, xdim)
            if xdtype != torch.bfloat16:
                dx = dx.to(xdtype)
        dw = tg.wgrad(dyr, xr)                               # (Np, K) fp32
        dw = dw[:N, :Kw].reshape(weight.shape)
        db = None
        if has_bias:   # column sums of the gradient rows (torch's strided reduction over 8 of them took 0.28 ms at 524288 rows)
            db = tg.colsum(dyr)[:N] if Np <= 512 else dyr[:, :N].sum(0, dtype=torch.float32)
        return dx, dw, db, None, None, None, None


class _BnReluRowsGemm(Function):
    """layer(relu(bn(x))) for x = the raw bf16 rows a _RowsGemm just produced together with their column sums: the
    BatchNorm is finalized from those sums (pdm_bn_finalize_stats) and applied, with the ReLU, WHILE the next contraction
    reads x (pdm_tg_gemm_nt / pdm_tg_wgrad with x_bn_coef) — the normalised tensor is never written or read; the values
    are bit for bit those of the separate operator.  Backward: data gradient of the layer -> BatchNorm + ReLU backward
    (pdm_bn_relu_backward over x and that gradient) -> gradient of x, dgamma, dbeta; weight gradient with the layer's
    input recomputed on the fly.  gamma / beta / running statistics arrive at x's (possibly zero-padded) width."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, stats, gamma, beta, running_mean, running_var, eps, momentum, weight, bias, want_stats, keep_pad,
                in_link=None, out_link=None, pool_ns=0):
        from . import train_gemm as tg
        ctx.in_link, ctx.link = in_link, out_link
        ctx.set_materialize_grads(False)
        xr = tg.row_view(x)
        R, K = xr.shape
        N = weight.shape[0]
        w2 = weight.reshape(N, -1)
        Kw, Np = w2.shape[1], _round8(N)
        assert xr.dtype == torch.bfloat16 and Kw <= K < Kw + 8 and K % 8 == 0 and (stats is None or stats.shape[1:] == (K, 2))
        coef = torch.empty((4, K), dtype=torch.float32, device=x.device)
        if stats is None:
            # x does not come from a contraction (the heat-map head's depthwise output): the statistics pass of the operator on its own,
            # then BatchNorm + ReLU ride in this contraction's load path like anywhere else — no normalised tensor
            assert xr.stride(0) == K
            scratch = torch.empty((_native.lib().pdm_bn_parts(0, R, K, 1), K, 2), dtype=torch.float32, device=x.device)
            _native.call("pdm_bn_forward_coef", torch.cuda.current_stream(x.device).cuda_stream, 1, R, K, xr.data_ptr(), gamma.data_ptr(),
                         beta.data_ptr(), float(eps), float(momentum), running_mean.data_ptr(), running_var.data_ptr(), coef.data_ptr(),
                         scratch.data_ptr())
        else:
            _native.call("pdm_bn_finalize_stats", torch.cuda.current_stream(x.device).cuda_stream, R, K, gamma.data_ptr(), beta.data_ptr(),
                         float(eps), float(momentum), running_mean.data_ptr(), running_var.data_ptr(), coef.data_ptr(), stats.data_ptr(),
                         stats.shape[0])
        wb, wt = tg.pack_weight_pair(w2, Np, K)
        if bias is not None and Np != N:
            bias = torch.cat([bias.detach().float(), bias.new_zeros(Np - N, dtype=torch.float32)])
        if pool_ns and want_stats and out_link is not None:
            y, st, out_link['pool'] = tg.gemm_nt(xr, wb, bias=bias, stats=True, x_bn_coef=coef, pool_ns=pool_ns)
        elif want_stats:
            y, st = tg.gemm_nt(xr, wb, bias=bias, stats=True, x_bn_coef=coef)
        else:
            y, st = tg.gemm_nt(xr, wb, bias=bias, x_bn_coef=coef), None
        ctx.save_for_backward(xr, coef, weight, wt)
        ctx.geom = (tuple(x.shape), x.dim(), N, Np, K, Kw, bias is not None)
        out = _rows_to_layout(y, x, Np if keep_pad else N)
        if st is not None:
            ctx.mark_non_differentiable(st)
        return out, st

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy, _dstats=None):
        from . import train_gemm as tg
        if dy is None:
            return (None,) * 15
        xr, coef, weight, wt = ctx.saved_tensors
        xshape, xdim, N, Np, K, Kw, has_bias = ctx.geom
        R = xr.shape[0]
        dyr = tg.row_view(dy)
        if dyr is None or dyr.dtype != torch.bfloat16 or dyr.shape[1] != Np or dyr.stride(0) % 8:
            nc = dy.shape[1]
            src = dy.movedim(1, -1).reshape(R, nc) if xdim > 2 else dy
            dyr = torch.zeros((R, Np), dtype=torch.bfloat16, device=dy.device) if Np != nc else torch.empty((R, Np), dtype=torch.bfloat16, device=dy.device)
            dyr[:, :nc].copy_(src)
        # the gradient statistics of the BatchNorm in front (over da and x) come out of the contraction that forms da
        bs = (xr, coef) if BWD_STATS_IN_GEMM and xr.stride(0) % 8 == 0 else None
        dyr, da, partial = _take_lazy_bn_backward(ctx.link, dyr, wt, True, bs)   # this layer's own output gradient may arrive unformed (see there)
        if da is None:                                       # gradient of relu(bn(x)), (R, K) bf16
            da, partial = tg.gemm_nt_bs(dyr, wt, xr, coef) if bs is not None else (tg.gemm_nt(dyr, wt), None)
        dw = tg.wgrad(dyr, xr, x_bn_coef=coef)[:N, :Kw].reshape(weight.shape)   # the layer's input recomputed while it is read
        db = None
        if has_bias:
            db = tg.colsum(dyr)[:N] if Np <= 512 else dyr[:, :N].sum(0, dtype=torch.float32)
        stream = torch.cuda.current_stream(dy.device).cuda_stream
        if partial is not None:
            grads = tg.bn_bwd_finalize(R, coef, partial)
        else:
            grads = torch.empty((4, K), dtype=torch.float32, device=dy.device)
            parts = _native.lib().pdm_bn_parts(0, R, K, 1)
            scratch = torch.empty((parts, K, 2), dtype=torch.float32, device=dy.device)
            _native.call("pdm_bn_relu_backward_stats", stream, 1, 0, R, K, 1, xr.data_ptr(), da.data_ptr(), coef.data_ptr(),
                         grads.data_ptr(), scratch.data_ptr(), 1)
        if LAZY_BN_BACKWARD and ctx.in_link is not None:
            # The BatchNorm + ReLU backward's elementwise half is left to the producer of x: its data gradient forms
            # dx = scale (da [bn(x) > 0] - p - (x - mean) q) while it reads da and x (pdm_tg_gemm_nt_dy) and writes it out for its
            # weight gradient.  What travels back through autograd is `da` itself; the link says how to read it.
            ctx.in_link['lazy'] = (xr, coef, grads, da.data_ptr())
            dx = da
        else:
            dx = torch.empty_like(xr)
            _native.call("pdm_bn_relu_backward_apply", stream, 1, 0, R, K, 1, xr.data_ptr(), da.data_ptr(), dx.data_ptr(), coef.data_ptr(),
                         grads.data_ptr(), 1)
        return (_rows_to_layout(dx, None, K, xshape, xdim), None, grads[0], grads[1], None, None, None, None, dw, db, None, None,
                None, None, None)


_identity_coef = {}


def _relu_coef(K, device):
    """(coef, grads) that make the BatchNorm + ReLU forms of the contractions a plain ReLU: mean 0, invstd 1, scale 1, shift 0 and
    p = q = 0 — relu((x - 0) 1 + 0) = relu(x) and 1 (g [x > 0] - 0 - x 0) = g [x > 0], exactly."""
    key = (K, device)
    if key not in _identity_coef:
        coef = torch.zeros((4, K), dtype=torch.float32, device=device)
        coef[1].fill_(1.0); coef[2].fill_(1.0)
        _identity_coef[key] = (coef, torch.zeros((4, K), dtype=torch.float32, device=device))
    return _identity_coef[key]


class _ReluRowsGemm(Function):
    """layer(relu(x)) for x = the raw bf16 rows a _RowsGemm just produced (Conv -> ReLU -> Conv without a BatchNorm: the heat-map
    head's output stack): the ReLU rides in the second contraction's load path and its backward in the first one's data gradient,
    through the BatchNorm + ReLU forms of the kernels with identity coefficients (_relu_coef) — relu(x) is never written, its
    gradient mask never applied in a pass of its own (72 M elements at bs = 32: ~0.13 ms of a step as torch kernels)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, want_stats, keep_pad, in_link=None, out_link=None):
        from . import train_gemm as tg
        ctx.in_link, ctx.link = in_link, out_link
        ctx.set_materialize_grads(False)
        xr = tg.row_view(x)
        R, K = xr.shape
        N = weight.shape[0]
        w2 = weight.reshape(N, -1)
        Kw, Np = w2.shape[1], _round8(N)
        assert xr.dtype == torch.bfloat16 and Kw <= K < Kw + 8 and K % 8 == 0
        coef, _ = _relu_coef(K, x.device)
        wb, wt = tg.pack_weight_pair(w2, Np, K)
        if bias is not None and Np != N:
            bias = torch.cat([bias.detach().float(), bias.new_zeros(Np - N, dtype=torch.float32)])
        if want_stats:
            y, st = tg.gemm_nt(xr, wb, bias=bias, stats=True, x_bn_coef=coef)
        else:
            y, st = tg.gemm_nt(xr, wb, bias=bias, x_bn_coef=coef), None
        ctx.save_for_backward(xr, weight, wt)
        ctx.geom = (tuple(x.shape), x.dim(), N, Np, K, Kw, bias is not None)
        out = _rows_to_layout(y, x, Np if keep_pad else N)
        if st is not None:
            ctx.mark_non_differentiable(st)
        return out, st

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy, _dstats=None):
        from . import train_gemm as tg
        if dy is None:
            return (None,) * 7
        xr, weight, wt = ctx.saved_tensors
        xshape, xdim, N, Np, K, Kw, has_bias = ctx.geom
        R = xr.shape[0]
        coef, grads = _relu_coef(K, xr.device)
        dyr = tg.row_view(dy)
        if dyr is None or dyr.dtype != torch.bfloat16 or dyr.shape[1] != Np or dyr.stride(0) % 8:
            nc = dy.shape[1]
            src = dy.movedim(1, -1).reshape(R, nc) if xdim > 2 else dy
            dyr = torch.zeros((R, Np), dtype=torch.bfloat16, device=dy.device) if Np != nc else torch.empty((R, Np), dtype=torch.bfloat16, device=dy.device)
            dyr[:, :nc].copy_(src)
        dyr, da, _ = _take_lazy_bn_backward(ctx.link, dyr, wt, True)
        if da is None:
            da = tg.gemm_nt(dyr, wt)                         # gradient of relu(x), (R, K) bf16
        dw = tg.wgrad(dyr, xr, x_bn_coef=coef)[:N, :Kw].reshape(weight.shape)
        db = None
        if has_bias:
            db = tg.colsum(dyr)[:N] if Np <= 512 else dyr[:, :N].sum(0, dtype=torch.float32)
        if LAZY_BN_BACKWARD and ctx.in_link is not None:
            ctx.in_link['lazy'] = (xr, coef, grads, da.data_ptr())     # the producer's data gradient applies the mask while it reads da and x
            dx = da
        else:
            dx = torch.empty_like(xr)
            _native.call("pdm_bn_relu_backward_apply", torch.cuda.current_stream(dy.device).cuda_stream, 1, 0, R, K, 1, xr.data_ptr(),
                         da.data_ptr(), dx.data_ptr(), coef.data_ptr(), grads.data_ptr(), 1)
        return _rows_to_layout(dx, None, K, xshape, xdim), dw, db, None, None, None, None


def relu_rows_linear(x, layer, want_stats=False, keep_pad=False, in_link=None, out_link=None):
    """layer(relu(x)) through _ReluRowsGemm for x fresh out of a rows contraction, or (None, None) when the form does not apply."""
    from . import train_gemm as tg
    if not (ENABLED and ROWS_GEMM and BN_IN_GEMM and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] <= 512 and _bf16_autocast()
            and layer.weight.dtype == torch.float32 and x.dim() in (2, 3, 4) and in_link is not None):
        return None, None
    if isinstance(layer, nn.Linear):
        kin = layer.in_features
    else:
        if not (all(k == 1 for k in layer.kernel_size) and all(v == 1 for v in layer.stride) and all(v == 0 for v in layer.padding)
                and all(v == 1 for v in layer.dilation) and layer.groups == 1 and isinstance(layer.padding, tuple)
                and x.dim() == layer.weight.dim()):
            return None, None
        kin = layer.in_channels
    K = x.shape[1]
    xr = tg.row_view(x)
    if K != _round8(kin) or K % 8 or xr is None or xr.stride(0) != K:
        return None, None
    return _ReluRowsGemm.apply(x, layer.weight, layer.bias, bool(want_stats), bool(keep_pad), in_link, out_link)


def bn_rows_linear(x, stats, bn, layer, want_stats=False, keep_pad=False, in_link=None, out_link=None, pool_ns=0):
    """layer(relu(bn(x))) through _BnReluRowsGemm, or (None, None) when the form does not apply (the caller then runs the
    BatchNorm operator and the layer one after the other)."""
    from . import train_gemm as tg
    if not (ENABLED and ROWS_GEMM and BN_IN_GEMM and x.is_cuda and x.shape[1] <= 512 and _bf16_autocast() and layer.weight.dtype == torch.float32
            and x.dtype == torch.bfloat16 and x.dim() in (2, 3, 4) and (applies(x, bn) or _padded_applies(x, bn))):
        return None, None
    if stats is None and (not BN_FROM_X or x.shape[1] != bn.num_features):     # (no statistics from a producer: see BN_FROM_X)
        return None, None
    if isinstance(layer, nn.Linear):
        kin = layer.in_features
    else:
        if not (all(k == 1 for k in layer.kernel_size) and all(v == 1 for v in layer.stride) and all(v == 0 for v in layer.padding)
                and all(v == 1 for v in layer.dilation) and layer.groups == 1 and isinstance(layer.padding, tuple)
                and x.dim() == layer.weight.dim()):
            return None, None
        kin = layer.in_channels
    C, K = bn.num_features, x.shape[1]
    xr = tg.row_view(x)
    if kin != C or K != _round8(C) or xr is None or xr.stride(0) != K or (stats is not None and stats.shape[1] != K) or _layout(x) is None \
            or _layout(x)[0] != 0:
        return None, None
    gamma, beta, rm, rv = bn.weight, bn.bias, bn.running_mean, bn.running_var
    if K != C:   # zero-padded width: zero gamma / beta on the padding (it stays zero), temporary running statistics
        z = bn.weight.new_zeros(K - C)
        gamma, beta, rm, rv = torch.cat([gamma, z]), torch.cat([beta, z]), torch.cat([rm, z]), torch.cat([rv, z + 1.0])
    out = _BnReluRowsGemm.apply(x, stats, gamma, beta, rm, rv, bn.eps, bn.momentum, layer.weight, layer.bias, bool(want_stats), bool(keep_pad),
                                in_link, out_link, int(pool_ns))
    with torch.no_grad():
        if K != C:
            bn.running_mean.copy_(rm[:C]); bn.running_var.copy_(rv[:C])
        _bump(bn)
    return out


# 1 (default): a BatchNorm + ReLU whose bf16 input does NOT come from a rows contraction (the heat-map head's first one, behind the
# depthwise convolution) still rides in the next contraction's load path: its statistics pass runs alone (pdm_bn_forward_coef), the
# apply pass and the normalised tensor (288 MB at bs = 32) disappear, and its gradient statistics come out of that contraction's data
# gradient like everywhere else.
BN_FROM_X = os.environ.get("PDM_BN_FROM_X", "1") == "1"

# 1 (default): BatchNorm + ReLU of an inner layer ride in the next contraction's load path (_BnReluRowsGemm; bit-identical results,
# the normalised tensor is never stored).  Measured at bs = 32, A/B twice: 26.17 / 26.24 ms per step with, 26.79 / 26.87
# without.  (An earlier A/B read 29.4-29.9 against 29.3-29.4 and kept the separate operator: the step was bound by the HOST
# then — a pageable host-to-device copy and two boolean-mask indexings in the target assignment stalled the issue thread
# every step — so device-side savings did not show.)
BN_IN_GEMM = os.environ.get("PDM_BN_IN_GEMM", "1") == "1"


def _rows_to_layout(rows, like, channels, shape=None, dim=None):
    """(R, ld) row storage -> the logical tensor of `like`'s layout with `channels` channels (a view)."""
    shape = tuple(like.shape) if shape is None else shape
    dim = like.dim() if dim is None else dim
    if dim == 2:
        return rows[:, :channels]
    lead = (shape[0],) + tuple(shape[2:])
    return rows.view(*lead, rows.shape[1])[..., :channels].movedim(-1, 1)


def rows_linear(x, layer, want_stats=False, keep_pad=False, link=None, pool_ns=0):
    """layer(x) for a 1x1 convolution / Linear through _RowsGemm when x is (castable to) bf16 rows on the GPU under bf16
    autocast; returns (y, stats) — stats None when not requested or not taken; (None, None) when the form does not apply.
    keep_pad: y keeps round8(out_channels) channels (the extra ones zero) for a BatchNorm over the padded width."""
    from . import train_gemm as tg
    if not (ENABLED and ROWS_GEMM and x.is_cuda and _bf16_autocast() and layer.weight.dtype == torch.float32 and x.dim() in (2, 3, 4)):
        return None, None
    if isinstance(layer, nn.Linear):
        kin = layer.in_features
    else:
        if not (all(k == 1 for k in layer.kernel_size) and all(v == 1 for v in layer.stride) and all(v == 0 for v in layer.padding)
                and all(v == 1 for v in layer.dilation) and layer.groups == 1 and isinstance(layer.padding, tuple)
                and x.dim() == layer.weight.dim()):
            return None, None
        kin = layer.in_channels
    K = x.shape[1]
    if x.dtype not in (torch.bfloat16, torch.float32) or K != _round8(kin) or tg.row_view(x) is None:
        return None, None
    return _RowsGemm.apply(x, layer.weight, layer.bias, bool(want_stats), bool(keep_pad), link, int(pool_ns))


ROWS_GEMM = os.environ.get("PDM_ROWS_GEMM", "1") != "0"   # 0: the round-2 path (vendor GEMMs) for A/B measurements


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


class _BNMeta(type):
    """isinstance(m, _BN): BatchNorm1d / BatchNorm2d, and SyncBatchNorm while it has nothing to synchronise with (no process
    group, or a single rank) — what tools/train.py:130-131 of the reference turns every BatchNorm into under --sync_bn.  With
    several ranks a SyncBatchNorm keeps torch's own implementation (its statistics cross ranks: an all-gather this operator
    does not do); the layers around it still run on the rows kernels."""

    def __instancecheck__(cls, m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            return True
        if isinstance(m, nn.SyncBatchNorm):
            import torch.distributed as dist
            return not (dist.is_available() and dist.is_initialized() and dist.get_world_size(m.process_group) > 1)
        return False


class _BN(metaclass=_BNMeta):
    pass


def _stats_wanted(mods, i):
    """for the Linear / convolution at mods[i]: (take BatchNorm column sums in its epilogue?, keep a zero-padded output width?)"""
    nxt = mods[i + 1] if i + 1 < len(mods) else None
    want = isinstance(nxt, _BN) and nxt.training and _round8(nxt.num_features) // 8 <= 256
    # an odd width (196) travels zero-padded to a multiple of 8 through its BatchNorm into the next layer
    pad = want and nxt.num_features % 8 != 0 and i + 3 < len(mods) and \
        (isinstance(mods[i + 3], nn.Linear) or type(mods[i + 3]) in (nn.Conv1d, nn.Conv2d)) and isinstance(mods[i + 2], nn.ReLU)
    return want and (nxt.num_features % 8 == 0 or pad), pad


class TrainSequential(nn.Sequential):
    """nn.Sequential whose (BatchNorm, ReLU) pairs run fused in training mode on the GPU (see the module docstring)."""

    def forward_max_pooled(self, x):
        """F.max_pool2d(self(x), kernel_size=[1, x.size(3)]) for a stack that ends in (BatchNorm2d, ReLU): in training
        mode on the GPU the last pair and the pooling run as one operator (the normalised tensor is never written)."""
        mods = list(self)
        if (ENABLED and self.training and x.is_cuda and len(mods) >= 3 and isinstance(mods[-1], nn.ReLU)
                and isinstance(mods[-2], _BN) and mods[-2].weight.dim() == 1):
            ns = x.shape[3] if x.dim() == 4 else 0
            bn = mods[-2]
            # the last contraction may leave the pooled operator's statistics in its epilogue: groups of ns consecutive rows
            tail = {'pool_ns': ns} if (POOL_IN_GEMM and 4 <= ns <= 128 and ns & (ns - 1) == 0 and bn.num_features % 8 == 0) else None
            h = self._run(x, mods[:-2], tail)
            if pool_applies(h, bn):
                with torch.no_grad():
                    _bump(bn)
                keeps = tail['link'].get('pool') if tail is not None and tail.get('link') is not None else None
                if keeps is not None and tail.get('stats') is not None and h.dtype == torch.bfloat16:
                    return _BnReluPool.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, tail['stats'],
                                             keeps[0], keeps[1])
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
    def _run(x, mods, tail=None):
        if _bump_later and _bump_later[-1] is not None:      # inside a counter_scope (or an outer stack): that one collects
            return TrainSequential._run_stack(x, mods, tail)
        _bump_later.append([])
        try:
            return TrainSequential._run_stack(x, mods, tail)
        finally:
            counters = _bump_later.pop()
            if counters:
                with torch.no_grad():
                    torch._foreach_add_(counters, 1)

    @staticmethod
    def _run_stack(x, mods, tail=None):
        """tail: None, or a dict {'pool_ns': ns} of forward_max_pooled — the LAST contraction of `mods` is asked for the pooled operator's
        statistics (rows_linear / bn_rows_linear with pool_ns), and the stack's final (stats, link) are handed back in it."""
        pool_ns = tail['pool_ns'] if tail is not None else 0
        i = 0
        stats = None      # column sums of x taken by the GEMM that produced it, for the BatchNorm right behind it
        link = None       # shared with the autograd node that produced x (a rows GEMM): see _take_lazy_bn_backward
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _BN) and (applies(x, m) or _padded_applies(x, m)):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                nxt = mods[i + 2] if relu and i + 2 < len(mods) else None
                if (stats is not None or (BN_FROM_X and x.dtype == torch.bfloat16)) and nxt is not None and \
                        (isinstance(nxt, nn.Linear) or type(nxt) in (nn.Conv1d, nn.Conv2d)):
                    # Conv -> BN -> ReLU -> Conv: the BatchNorm + ReLU ride in the second contraction's load path
                    want, pad = _stats_wanted(mods, i + 2)
                    out_link = {}
                    last = i + 3 == len(mods)
                    y, st = bn_rows_linear(x, stats, m, nxt, want or (last and pool_ns > 0), pad, link, out_link, pool_ns if last else 0)
                    if y is not None:
                        x, stats, link = y, st, out_link
                        i += 3
                        continue
                # an fp32 map (the heat-map head's depthwise output) that a bf16 contraction reads next: the BatchNorm writes bf16
                to_bf16 = (x.dtype == torch.float32 and ROWS_GEMM and nxt is not None and _bf16_autocast()
                           and (isinstance(nxt, nn.Linear) or (type(nxt) in (nn.Conv1d, nn.Conv2d) and all(k == 1 for k in nxt.kernel_size)
                                                               and nxt.groups == 1)))
                # (stats is not None: x is the direct output of a rows GEMM node, which shares `link` with this operator)
                x = batch_norm_relu(x, m, relu, stats, out_bf16=to_bf16, link=link if stats is not None else None)
                stats = link = None
                i += 2 if relu else 1
                continue
            if isinstance(m, nn.ReLU) and link is not None and i + 1 < len(mods) and \
                    (isinstance(mods[i + 1], nn.Linear) or type(mods[i + 1]) in (nn.Conv1d, nn.Conv2d)):
                # Conv -> ReLU -> Conv: the ReLU rides in the second contraction's load path (and its backward in the first's data gradient)
                want, pad = _stats_wanted(mods, i + 1)
                out_link = {}
                y, st = relu_rows_linear(x, mods[i + 1], want, pad, link, out_link)
                if y is not None:
                    x, stats, link = y, st, out_link
                    i += 2
                    continue
            stats = link = None
            if isinstance(m, nn.Linear) or type(m) in (nn.Conv1d, nn.Conv2d):
                want, pad = _stats_wanted(mods, i)
                new_link = {}
                last = i + 1 == len(mods)
                y, st = rows_linear(x, m, want or (last and pool_ns > 0), pad, new_link, pool_ns if last else 0)
                if y is not None:
                    x, stats, link = y, st, new_link
                else:
                    kin = m.in_features if isinstance(m, nn.Linear) else m.in_channels
                    if x.shape[1] != kin and x.shape[1] == _round8(kin):
                        x = x[:, :kin]               # zero-padded channels (see _RowsGemm) the torch layer does not know about
                    x = tall_linear(x, m) if isinstance(m, nn.Linear) else conv1x1(x, m)
                i += 1
            else:
                x = m(x)
                i += 1
        if tail is not None:
            tail['stats'], tail['link'] = stats, link
        return x
