#!/usr/bin/env python3
"""Fused BatchNorm+ReLU (csrc/bn_relu.hip) and torch's own GPU batch_norm + relu against an fp64 CPU reference on
the same inputs: mean absolute and mean signed error of y and dx, relative error of dgamma / dbeta.
    python tools/diag/bn_error_stats.py
"""
import copy, os, sys
import torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pdm_ssd_amd import fused_bn

dev = torch.device("cuda:0")


def run(shape, cl, dtype, mean_shift):
    torch.manual_seed(0)
    C = shape[1]
    bn = (nn.BatchNorm1d if len(shape) <= 3 else nn.BatchNorm2d)(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5); bn.bias.copy_(torch.randn(C) * 0.2)
    x = (torch.randn(shape) * 1.5 + mean_shift * torch.randn(1, C, *([1] * (len(shape) - 2)))).to(dtype)
    gy = torch.randn(shape).to(dtype)
    # fp64 truth from the same (rounded) inputs
    b64 = copy.deepcopy(bn).double().train()
    x64 = x.double().requires_grad_(True)
    y64 = torch.relu(b64(x64)); y64.backward(gy.double())
    out = {}
    for name in ("fused", "torch"):
        b = copy.deepcopy(bn).to(dev).train()
        xg = x.to(dev)
        if cl:
            xg = xg.contiguous(memory_format=torch.channels_last)
        xg.requires_grad_(True)
        if name == "fused":
            y = fused_bn.batch_norm_relu(xg, b, True)
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
                y = torch.relu(b(xg))
        y.backward(gy.to(dev).to(y.dtype))
        ey = (y.detach().double().cpu() - y64.detach())
        ex = (xg.grad.double().cpu() - x64.grad)
        out[name] = (y.dtype, ey.abs().mean().item(), ey.mean().item(), ex.abs().mean().item(), ex.mean().item(),
                     ((b.weight.grad.double().cpu() - b64.weight.grad).norm() / b64.weight.grad.norm()).item(),
                     ((b.bias.grad.double().cpu() - b64.bias.grad).norm() / b64.bias.grad.norm()).item())
    print(f"shape {shape} cl={cl} {dtype} mean_shift={mean_shift}")
    for k, v in out.items():
        print(f"   {k:6s} y:{str(v[0]):15s} |ey|={v[1]:.3e} bias={v[2]:+.2e}  |edx|={v[3]:.3e} bias={v[4]:+.2e}  dgamma {v[5]:.2e} dbeta {v[6]:.2e}")


for dt in (torch.bfloat16, torch.float32):
    run((2, 16, 512, 16), True, dt, 0.0)
    run((2, 16, 512, 16), True, dt, 5.0)
    run((8192, 256), False, dt, 1.0)
    run((2, 128, 2048, 1), False, dt, 1.0)
