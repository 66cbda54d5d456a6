#!/usr/bin/env python3
"""pdm_fp_mlp_fused_pre at the bench shapes of FP1 / FP2 (bs=32): the register-resident chain (rows_chain.hip
fp_chain_kernel) against the LDS-tiled forms it replaces, same inputs, HIP-event timing, max abs difference.
    python tools/diag/fp_chain_rate.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pdm_ssd_amd import _native, fused  # noqa: E402


def real_interp(B, n, m, dev):
    """Three-NN indices / weights as the backbone computes them: FPS-sampled known set of a synthetic cloud (spatially local
    gathers, unlike random indices)."""
    from pdm_ssd_amd import synthetic
    from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
    from pdm_ssd_amd.pointnet2_batch.pointnet2_modules import PointnetFPModule
    xyz = torch.from_numpy(np.ascontiguousarray(synthetic.uniform_clouds(B, 16384, 1234)[:, :, :3])).to(dev)
    while xyz.shape[1] > n:     # FPS chain down to the unknown set's size
        i = pu.furthest_point_sample(xyz, xyz.shape[1] // 4)
        xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), i).transpose(1, 2).contiguous()
    i = pu.furthest_point_sample(xyz, m)
    known = pu.gather_operation(xyz.transpose(1, 2).contiguous(), i).transpose(1, 2).contiguous()
    idx, w = PointnetFPModule.interpolation(xyz, known)
    return idx.contiguous(), w.contiguous()


def run(name, B, n, m, cs, c1, c2, dev):
    torch.manual_seed(0)
    conv1 = torch.nn.Conv2d(max(cs, 1), c1, 1, bias=False); bn1 = torch.nn.BatchNorm2d(c1)
    conv2 = torch.nn.Conv2d(c1, c2, 1, bias=False); bn2 = torch.nn.BatchNorm2d(c2)
    pk = fused.PackedMLP([(conv1, bn1.eval()), (conv2, bn2.eval())], dev, min_in=16)
    z = torch.randn(B, m, c1, device=dev)
    skip = torch.randn(B, n, max(cs, 1), device=dev)
    idx = torch.randint(0, m, (B, n, 3), device=dev, dtype=torch.int32)
    w = torch.rand(B, n, 3, device=dev); w = (w / w.sum(-1, keepdim=True)).contiguous()
    if "--real" in sys.argv:
        idx, w = real_interp(B, n, m, dev)
    outs = {}
    for chain in ((1, 2) if "--nt" in sys.argv else (0, 1)):
        _native.lib().pdm_tune_fp_chain_nt(1 if chain == 2 else 0)
        _native.lib().pdm_tune_fused_chain(1 if chain else 0)
        out = torch.zeros(B, n, c2, device=dev)
        for _ in range(150):   # the clock settles over the first ~100 ms of load
            fused.fp_forward_pre(pk, z, skip, idx, w, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fused.fp_forward_pre(pk, z, skip, idx, w, out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        gf = 2.0 * B * n * (pk.dims[0] * c1 + c1 * c2) / 1e9
        print(f"{name} chain={chain}: {us:8.1f} us  {gf / us * 1e-3 * 1e3:6.1f} TFLOP/s (padded skip width {pk.dims[0]})", flush=True)
        outs[chain] = out
    _native.lib().pdm_tune_fused_chain(1); _native.lib().pdm_tune_fp_chain_nt(0)
    ks = sorted(outs)
    print(f"{name} max |form {ks[0]} - form {ks[1]}| = {float((outs[ks[0]] - outs[ks[1]]).abs().max()):.3e}  (0 tiled, 1 chain, 2 chain with non-temporal stores)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--lib":    # a timing build from `make -C pdm_ssd_amd/csrc diag-fpc`
        _native.LIB_PATH = os.path.abspath(sys.argv[2])
        print("library:", _native.LIB_PATH, flush=True)
    dev = torch.device("cuda:0")
    if "--one-wg-per-cu" in sys.argv:
        _native.lib().pdm_tune_fp_chain_pad_lds(90 * 1024)
    run("FP1 (1 -> 128 -> 128, 524288 rows)", 32, 16384, 4096, 1, 128, 128, dev)
    run("FP2 (96 -> 256 -> 256, 131072 rows)", 32, 4096, 1024, 96, 256, 256, dev)
