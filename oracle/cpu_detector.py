"""CPU statement of ONE TRAINING STEP of the whole PDM-SSD detector (PointNet2MSG backbone + PDM neck + hybrid head with
its real losses), fp32 or bf16-emulating.

TEST INFRASTRUCTURE ONLY (same rule as cpu_oracle.py / cpu_backbone.py / cpu_autograd.py): the checker of BASELINE
configs[3] in tests/test_configs_gpu.py.  Nothing in pdm_ssd_amd/ imports it.

What it follows: the detector loop and loss sum of /root/reference/pcdet/models/detectors/point_rcnn.py:13-30, the
point head's targets and losses of dense_heads/point_head_template.py:82-89,127-183 (through this repo's own
PointHeadBox code on the CPU, with the oracle's points_in_boxes standing in for the HIP kernel), the heat-map focal
loss of utils/loss_utils.py:335-345, and cpu_autograd.py for the backbone and the neck (oracle operators).

Two modes:
* bf16=False — the model's OWN torch layers on the CPU in fp32 (Conv/Linear/BatchNorm/ReLU/max-pool of torch) over the
  oracle operators.
* bf16=True — a bf16-EMULATING graph: fp32 arithmetic on the CPU with a round-to-nearest-even to bf16 at every point
  where the GPU training path under `torch.autocast(bfloat16)` holds a bf16 tensor, forward AND backward:
    - 1x1 convolutions / Linear layers: operands rounded to bf16, fp32 accumulation, output rounded to bf16; in the
      backward the incoming gradient is bf16, the input gradient is rounded to bf16, the weight / bias gradients stay
      fp32 (the split-K forms of fused_bn.py hand fp32 sums to the fp32 parameters);
    - BatchNorm(train) + ReLU: fp32 statistics over the bf16 input, y = relu((x - mean) * (gamma * invstd) + beta)
      rounded to bf16; backward dx = scale * (g - p - (x - mean) q) rounded to bf16, g = dy where the pre-ReLU value
      is > 0, dgamma / dbeta fp32 (the arithmetic of csrc/bn_relu.hip); on an fp32 input nothing is rounded;
    - the SA scales' last BatchNorm + ReLU + max over nsample: the pooled element is the FIRST neighbour attaining the
      max of the layer input x (min where gamma < 0) — pdm_bn_relu_pool_forward's rule;
    - QueryAndGroup output rounded to bf16 (pdm_group_concat_cl), three_interpolate / PDM scatter / losses in fp32 (their
      autograd Functions cast to fp32); the heat-map head's depthwise 3x3 computes in fp32 on the fp32 map and hands on a
      bf16 output (its gradient arrives in bf16).
  The comparison is then between two bf16 computations that differ only in fp32 summation order (and the bf16 roundings
  that a last-bit difference flips), instead of a bf16 result against an fp32 one.
"""
import copy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import cpu_autograd as ca
from . import cpu_oracle as o


ROUND = True   # False: the emulating graph with every rounding switched off (cross-check of its formulas against torch)


def bf16r(t):
    """round to nearest even to bf16, returned as fp32 (what v_cvt_pk_bf16_f32 / torch's cast do)"""
    return t.to(torch.bfloat16).to(torch.float32) if ROUND else t


class _AsBf16(Function):
    """marks a tensor that is bf16 on the GPU: value and gradient are both rounded"""

    @staticmethod
    def forward(ctx, x):
        return bf16r(x)

    @staticmethod
    def backward(ctx, g):
        return bf16r(g)


class _MatmulBf16(Function):
    """rows (R, Cin) x weight (Cout, Cin) [+ bias]: bf16 operands, fp32 accumulation, bf16 result"""

    @staticmethod
    def forward(ctx, x, w, b):
        xb, wb = bf16r(x), bf16r(w)
        ctx.save_for_backward(xb, wb)
        ctx.has_bias = b is not None
        y = xb @ wb.t()
        if b is not None:
            y = y + bf16r(b)
        return bf16r(y)

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        dy = bf16r(dy)
        dx = bf16r(dy @ wb)
        dw = dy.t() @ xb
        db = dy.sum(0) if ctx.has_bias else None
        return dx, dw, db


def _stats(x2, eps):
    """x2 (R, C) -> mean, invstd (fp32, from double sums as bn_finalize_fwd_kernel)"""
    xd = x2.double()
    mean = xd.mean(0)
    var = (xd - mean).square().mean(0).clamp(min=0.0)
    return mean.float(), (1.0 / torch.sqrt(var + eps)).float()


class _BnRelu(Function):
    """rows (R, C): train-mode BatchNorm [+ ReLU] with the arithmetic of csrc/bn_relu.hip; `rnd` rounds y / dx to bf16"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, rnd):
        mean, invstd = _stats(x, eps)
        scale = gamma * invstd
        d = x - mean
        pre = d * scale + beta
        y = torch.clamp(pre, min=0.0) if relu else pre
        ctx.save_for_backward(d, scale, invstd, pre)
        ctx.relu, ctx.rnd = relu, rnd
        return bf16r(y) if rnd else y

    @staticmethod
    def backward(ctx, dy):
        d, scale, invstd, pre = ctx.saved_tensors
        if ctx.rnd:
            dy = bf16r(dy)
        g = torch.where(pre > 0, dy, torch.zeros_like(dy)) if ctx.relu else dy
        count = d.shape[0]
        dbeta = g.double().sum(0)
        dgamma = (g * (d * invstd)).double().sum(0)
        p = (dbeta / count).float()
        q = (invstd.double() * dgamma / count).float()
        dx = scale * (g - p - d * q)
        return (bf16r(dx) if ctx.rnd else dx), dgamma.float(), dbeta.float(), None, None, None


class _BnReluPool(Function):
    """x (G, ns, C) -> (G, C): BatchNorm(train, statistics over all G * ns rows) + ReLU + max over ns as
    pdm_bn_relu_pool_forward / _backward do it (first neighbour attaining the extreme of x)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, rnd):
        G, ns, C = x.shape
        mean, invstd = _stats(x.reshape(-1, C), eps)
        scale = gamma * invstd
        up = scale >= 0
        imax = x.argmax(dim=1)                                   # first index attaining the maximum
        imin = x.argmin(dim=1)
        sel = torch.where(up[None, :], imax, imin)               # (G, C)
        xsel = x.gather(1, sel[:, None, :]).squeeze(1)
        pre = (xsel - mean) * scale + beta
        y = torch.clamp(pre, min=0.0)
        ctx.save_for_backward(x, mean, scale, invstd, pre, sel)
        ctx.rnd = rnd
        return bf16r(y) if rnd else y

    @staticmethod
    def backward(ctx, dy):
        x, mean, scale, invstd, pre, sel = ctx.saved_tensors
        G, ns, C = x.shape
        if ctx.rnd:
            dy = bf16r(dy)
        g = torch.where(pre > 0, dy, torch.zeros_like(dy))      # (G, C): the pooled gradient belongs to one element
        xsel = x.gather(1, sel[:, None, :]).squeeze(1)
        count = G * ns
        dbeta = g.double().sum(0)
        dgamma = (g * ((xsel - mean) * invstd)).double().sum(0)
        p = (dbeta / count).float()
        q = (invstd.double() * dgamma / count).float()
        gfull = torch.zeros_like(x).scatter_(1, sel[:, None, :], g[:, None, :])
        dx = scale * (gfull - p - (x - mean) * q)
        return (bf16r(dx) if ctx.rnd else dx), dgamma.float(), dbeta.float(), None, None


# ------------------------------------------------------------------------------------------------ layer interpreter

def _rows(x):
    """(B, C, ...) -> (rows, C) view/copy with the channel last, and the shape to go back"""
    xm = x.movedim(1, -1)
    return xm.reshape(-1, x.shape[1]), xm.shape


def _unrows(y2, shape_cl):
    return y2.reshape(*shape_cl[:-1], y2.shape[1]).movedim(-1, 1)


def _is_pointwise(m):
    return (isinstance(m, (nn.Conv1d, nn.Conv2d)) and all(k == 1 for k in m.kernel_size) and m.groups == 1) or isinstance(m, nn.Linear)


def emu_stack(mods, x, is_bf16, pooled=False):
    """Run a Conv/Linear -> BatchNorm -> ReLU ... stack the way the GPU training path does under bf16 autocast.
    x: (B, C, ...) or (rows, C) fp32 holding bf16 values when is_bf16.  pooled: the stack ends in (BatchNorm2d, ReLU)
    over (B, C, M, ns) and is followed by the max over ns (SA scales) -> returns (B, C, M).
    Returns (y, is_bf16)."""
    mods = list(mods)
    i = 0
    while i < len(mods):
        m = mods[i]
        if _is_pointwise(m):
            w = m.weight.reshape(m.weight.shape[0], -1)
            if x.dim() == 2:
                x = _MatmulBf16.apply(x, w, m.bias)
            else:
                x2, shp = _rows(x)
                x = _unrows(_MatmulBf16.apply(x2, w, m.bias), shp)
            is_bf16 = True
            i += 1
        elif isinstance(m, nn.Conv2d):                            # the heat-map head's depthwise 3x3: fp32 arithmetic on an fp32 map,
            # the OUTPUT (and the gradient that comes back for it) held in bf16 (pdm_bev_depthwise3x3_t)
            x = _AsBf16.apply(F.conv2d(x, m.weight, m.bias, m.stride, m.padding, m.dilation, m.groups))
            is_bf16 = True
            i += 1
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            last = i + (2 if relu else 1) >= len(mods)
            if pooled and last:
                assert relu and x.dim() == 4
                B, C, M, ns = x.shape
                y = _BnReluPool.apply(x.permute(0, 2, 3, 1).reshape(B * M, ns, C), m.weight, m.bias, m.eps, is_bf16)
                x = y.reshape(B, M, C).permute(0, 2, 1)
            elif x.dim() == 2:
                x = _BnRelu.apply(x, m.weight, m.bias, m.eps, relu, is_bf16)
            else:
                x2, shp = _rows(x)
                x = _unrows(_BnRelu.apply(x2, m.weight, m.bias, m.eps, relu, is_bf16), shp)
            i += 2 if relu else 1
        elif isinstance(m, nn.ReLU):
            x = torch.relu(x)                                     # exact on bf16 values
            i += 1
        else:
            raise NotImplementedError(type(m))
    return x, is_bf16


# ------------------------------------------------------------------------------------------------ the step

def _sa_forward_emu(sa, xyz, features):
    idx = o.furthest_point_sample(xyz, sa.npoint)
    new_xyz = np.ascontiguousarray(np.take_along_axis(xyz, idx[:, :, None].astype(np.int64), 1))
    outs = []
    for grouper, mlp in zip(sa.groupers, sa.mlps):
        g = _AsBf16.apply(ca._QueryAndGroup.apply(features, xyz, new_xyz, grouper.radius, grouper.nsample))
        y, _ = emu_stack(mlp, g, True, pooled=True)
        outs.append(y)
    return new_xyz, torch.cat(outs, dim=1)


def _fp_forward_emu(fp, unknown, known, unknown_feats, known_feats):
    dist, idx = o.three_nn(unknown, known)
    d = torch.from_numpy(dist)
    dist_recip = 1.0 / (d + 1e-8)
    weight = (dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)).numpy()
    interp = ca._ThreeInterpolate.apply(known_feats, idx, weight)              # fp32 operator (cast_inputs)
    x = interp if unknown_feats is None else torch.cat([interp, unknown_feats], dim=1)   # fp32 (autocast promotes)
    y, _ = emu_stack(fp.mlp, x.unsqueeze(-1), False)
    return y.squeeze(-1)


def _backbone_neck(model, clouds, bf16):
    bb, neck = model.backbone_3d, model.map_to_bev_module
    if not bf16:
        return ca.train_forward(bb, neck, clouds)
    xyz = np.ascontiguousarray(clouds[:, :, :3])
    feats = torch.from_numpy(np.ascontiguousarray(clouds[:, :, 3:].transpose(0, 2, 1))) if clouds.shape[2] > 3 else None
    l_xyz, l_feat = [xyz], [feats]
    for sa in bb.SA_modules:
        nx, nf = _sa_forward_emu(sa, l_xyz[-1], l_feat[-1])
        l_xyz.append(nx)
        l_feat.append(nf)
    sa_xyz, sa_feat = list(l_xyz), list(l_feat)
    for i in range(-1, -(len(bb.FP_modules) + 1), -1):
        l_feat[i - 1] = _fp_forward_emu(bb.FP_modules[i], l_xyz[i - 1], l_xyz[i], l_feat[i - 1], l_feat[i])
    pf = l_feat[0].permute(0, 2, 1).reshape(-1, l_feat[0].shape[1])
    out = {'point_features': pf, 'sa_xyz': sa_xyz, 'sa_features': sa_feat}
    if neck is not None:
        src = sa_feat[neck.source_layer]                                        # (B, Cin, P) bf16
        feat, _ = emu_stack(neck.proj, src, True)
        feat = feat.transpose(1, 2).contiguous()
        co, _ = emu_stack([neck.coef], src, True)
        co = co.transpose(1, 2)
        sh = co[..., :neck.nsh].contiguous()
        sigma = F.softplus(co[..., neck.nsh]) + neck.sigma_min                  # fp32 (autocast runs softplus in fp32)
        inv2s2 = (0.5 / (sigma * sigma)).contiguous()
        g = neck.grid
        spec = (g.origin, g.cell, g.inv_cell, (g.W, g.H, g.D), neck.dilation, neck.degree)
        grid, wsum = ca._PdmScatter.apply(feat, sh, inv2s2, sa_xyz[neck.source_layer], spec)
        if neck.normalize:
            B = grid.shape[0]
            w = wsum.unsqueeze(3)
            g5 = grid.view(B, g.H, g.W, neck.feature_dim, g.D)
            ok = w.abs() > 1e-6
            grid = torch.where(ok, g5 / torch.where(ok, w, torch.ones_like(w)), g5).reshape(B, g.H, g.W, -1)
        out['spatial_features'] = grid.permute(0, 3, 1, 2)
    return out


def detector_train_step(model, clouds, gt_boxes, bf16=False):
    """model: a PDMSSD on the CPU in train() mode (a private copy is made); clouds (B, N, 3 + C) numpy, gt_boxes
    (B, M, 8) numpy.  Runs forward + the detector's losses + backward on the CPU.  bf16: False = torch layers in fp32,
    True = the bf16-emulating graph, 'unrounded' = the emulating graph with its roundings switched off (must equal the
    torch-layer graph up to fp32 summation order: tests/test_cpu_autograd.py).
    -> {'loss': float, 'tb': {name: float}, 'grads': {parameter name: tensor}, 'point_features', 'spatial_features',
        'point_cls_preds', 'point_box_preds', 'hm_logits', 'point_cls_labels'}"""
    global ROUND
    model = copy.deepcopy(model).float().train()
    B = clouds.shape[0]
    ROUND = bf16 != 'unrounded'
    try:
        return _step(model, clouds, gt_boxes, bool(bf16), B)
    finally:
        ROUND = True


def _step(model, clouds, gt_boxes, bf16, B):
    from pdm_ssd_amd.dense_heads import point_head_template
    from pdm_ssd_amd import synthetic
    out = _backbone_neck(model, clouds, bf16)
    gt = torch.from_numpy(np.ascontiguousarray(gt_boxes))
    coords = torch.from_numpy(np.ascontiguousarray(synthetic.to_batch_points(clouds)[:, :4]))
    bd = {'batch_size': B, 'point_features': out['point_features'], 'point_coords': coords, 'gt_boxes': gt,
          'spatial_features': out['spatial_features'], 'points_per_sample_checked': True}
    ph, dh = model.point_head, model.dense_head

    def pib(points, boxes):   # the oracle answers where the product asks its HIP kernel
        return torch.from_numpy(o.points_in_boxes(points.detach().numpy(), boxes.detach().numpy()))
    saved = point_head_template.iou3d_nms_utils.points_in_boxes_gpu
    point_head_template.iou3d_nms_utils.points_in_boxes_gpu = pib
    try:
        if not bf16:
            bd = dh(bd)       # the modules' own torch layers on the CPU
            bd = ph(bd)
        else:
            targets = ph.assign_targets(bd)
            cls, _ = emu_stack(ph.cls_layers, out['point_features'], True)
            box, _ = emu_stack(ph.box_layers, out['point_features'], True)
            ph.forward_ret_dict = {'point_cls_preds': cls, 'point_box_preds': box,
                                   'point_cls_labels': targets['point_cls_labels'], 'point_box_labels': targets['point_box_labels']}
            x, isb = emu_stack(dh.shared_conv, out['spatial_features'], False)
            logits, _ = emu_stack(dh.hm, x, isb)
            dh.forward_ret_dict = {'hm_logits': logits, 'heatmap': dh.assign_targets(gt, logits.shape[2:])}
        loss, tb, _ = model.get_training_loss()
    finally:
        point_head_template.iou3d_nms_utils.points_in_boxes_gpu = saved
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    return {'loss': float(loss.detach()), 'tb': {k: float(v) for k, v in tb.items()}, 'grads': grads,
            'point_features': out['point_features'].detach(), 'spatial_features': out['spatial_features'].detach(),
            'sa_features': [None if t is None else t.detach() for t in out['sa_features']],
            'point_cls_preds': ph.forward_ret_dict['point_cls_preds'].detach(),
            'point_box_preds': ph.forward_ret_dict['point_box_preds'].detach(),
            'hm_logits': dh.forward_ret_dict['hm_logits'].detach(),
            'point_cls_labels': ph.forward_ret_dict['point_cls_labels'].detach()}
