#!/usr/bin/env python3
"""Stacked ball query (exhaustive wave-per-centre scan, csrc/stack_ops.hip): time and distance evaluations per second.
    python tools/diag/stack_bq_rate.py
Measured on MI355X: B=4 x 16384 points, 4096 centres each, r = 0.8: 135 us = 2.0e12 evaluations/s; B=8 x 65536,
r = 1.6: 570 us = 3.8e12/s.  A per-sample route through the batch operator's cell grid (one grid build + query + an
empty-ball marker per sample, one host read of the counts) was built, index-exact, and measured 369 us / 4455 us on
the same inputs — the batch grid kernels parallelise over clouds, so one cloud at a time serialises them — and removed.
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pdm_ssd_amd import synthetic
from pdm_ssd_amd.pointnet2_stack import pointnet2_stack_hip as ext, pointnet2_utils as su

dev = torch.device("cuda:0")
for B, n, m, r, ns in ((4, 16384, 4096, 0.8, 32), (8, 65536, 4096, 1.6, 32)):
    xyz = torch.from_numpy(np.concatenate([synthetic.lidar_like_clouds(1, n, 5 + b)[0, :, :3] for b in range(B)])).to(dev)
    new = torch.cat([xyz[b * n:(b + 1) * n][torch.randperm(n, device=dev)[:m]] for b in range(B)]).contiguous()
    xc = torch.full((B,), n, dtype=torch.int32, device=dev); nc = torch.full((B,), m, dtype=torch.int32, device=dev)
    out = {}
    for route in ("scan",):
        for _ in range(2):
            idx, _m = su.ball_query(r, ns, xyz, xc, new, nc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            idx, _m = su.ball_query(r, ns, xyz, xc, new, nc)
        e1.record(); torch.cuda.synchronize()
        out[route] = idx
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"B={B} n={n} m={m} r={r}: {route} {us:9.1f} us  {B * m * n / us * 1e6:.2e} evaluations/s", flush=True)
