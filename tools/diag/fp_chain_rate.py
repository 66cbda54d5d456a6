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


def run(name, B, n, m, cs, c1, c2, dev):
    torch.manual_seed(0)
    conv1 = torch.nn.Conv2d(max(cs, 1), c1, 1, bias=False); bn1 = torch.nn.BatchNorm2d(c1)
    conv2 = torch.nn.Conv2d(c1, c2, 1, bias=False); bn2 = torch.nn.BatchNorm2d(c2)
    pk = fused.PackedMLP([(conv1, bn1.eval()), (conv2, bn2.eval())], dev, min_in=16)
    z = torch.randn(B, m, c1, device=dev)
    skip = torch.randn(B, n, max(cs, 1), device=dev)
    idx = torch.randint(0, m, (B, n, 3), device=dev, dtype=torch.int32)
    w = torch.rand(B, n, 3, device=dev); w = (w / w.sum(-1, keepdim=True)).contiguous()
    outs = {}
    for chain in (0, 1):
        _native.lib().pdm_tune_fused_chain(chain)
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
    _native.lib().pdm_tune_fused_chain(1)
    print(f"{name} max |chain - tiled| = {float((outs[0] - outs[1]).abs().max()):.3e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--lib":    # a timing build from `make -C pdm_ssd_amd/csrc diag-fpc`
        _native.LIB_PATH = os.path.abspath(sys.argv[2])
        print("library:", _native.LIB_PATH, flush=True)
    dev = torch.device("cuda:0")
    if "--one-wg-per-cu" in sys.argv:
        _native.lib().pdm_tune_fp_chain_pad_lds(90 * 1024)
    run("FP1 (1 -> 128 -> 128, 524288 rows)", 32, 16384, 4096, 1, 128, 128, dev)
    run("FP2 (96 -> 256 -> 256, 131072 rows)", 32, 4096, 1024, 96, 256, 256, dev)
