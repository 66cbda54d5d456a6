#!/usr/bin/env python3
"""Per-operator timings at the PointNet2MSG shapes (HIP events on the launch stream).
Usage: python tools/microbench.py [--ops fps,bq,...] [--clouds uniform|lidar] [--batch 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pdm_ssd_amd import _native, synthetic
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="fps,bq,group,nn,interp")
    ap.add_argument("--clouds", default="uniform")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--model", action="store_true", help="per-call breakdown of one bench step")
    args = ap.parse_args()
    if args.model:
        model_breakdown(args.clouds, args.batch)
        return
    ops = args.ops.split(",")
    dev = torch.device("cuda:0")
    B = args.batch
    gen = synthetic.uniform_clouds if args.clouds == "uniform" else synthetic.lidar_like_clouds
    xyz0 = torch.from_numpy(np.ascontiguousarray(gen(B, 16384)[:, :, :3])).to(dev)
    levels = [xyz0]
    npts = [4096, 1024, 256, 64]
    for m in npts:
        idx = pu.furthest_point_sample(levels[-1], m)
        levels.append(pu.gather_operation(levels[-1].transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous())
    print(f"clouds={args.clouds} B={B}")
    if "fps" in ops:
        for variant in (0, 3, 1):
            _native.lib().pdm_tune_fps_variant(variant)
            for lv, m in enumerate(npts):
                x = levels[lv]
                ms = timeit(lambda: pu.furthest_point_sample(x, m), iters=5, warm=1)
                print(f"fps[v{variant}] N={x.shape[1]:6d} m={m:5d}: {ms:8.3f} ms  ({ms * 1e3 / max(m - 1, 1):.3f} us/iter)")
        _native.lib().pdm_tune_fps_variant(0)
    radii = [[0.1, 0.5], [0.5, 1.0], [1.0, 2.0], [2.0, 4.0]]
    chans = [1, 96, 256, 512]
    if "bq" in ops or "group" in ops:
        for lv in range(4):
            x, nx = levels[lv], levels[lv + 1]
            N, M = x.shape[1], nx.shape[1]
            feat = torch.randn(B, chans[lv], N, device=dev)
            for r, ns in zip(radii[lv], [16, 32]):
                idx = pu.ball_query(r, ns, x, nx)
                fill = float((idx != idx[:, :, :1]).float().mean())
                if "bq" in ops:
                    ms = timeit(lambda: pu.ball_query(r, ns, x, nx))
                    print(f"ball_query N={N:6d} M={M:5d} r={r} ns={ns}: {ms:8.3f} ms  (non-pad frac {fill:.2f})")
                if "group" in ops:
                    out = torch.empty(B, 3 + chans[lv], M, ns, device=dev)
                    ms = timeit(lambda: ext.query_and_group_wrapper(B, N, M, chans[lv], r, ns, x, nx, feat, idx, out))
                    by = B * (4 * M * ns + 12 * N + 4 * chans[lv] * N + 12 * M + 4 * (3 + chans[lv]) * M * ns)
                    ms_b = timeit(lambda: pu.ball_query(r, ns, x, nx))
                    print(f"group_concat C={chans[lv]:4d} M={M:5d} ns={ns}: {ms - ms_b:8.3f} ms  {by / 1e9 / ((ms - ms_b) / 1e3):8.1f} GB/s")
    if "nn" in ops or "interp" in ops:
        fc = [1024, 512, 512, 256]
        for lv in (3, 2, 1, 0):
            unk, kn = levels[lv], levels[lv + 1]
            n, m = unk.shape[1], kn.shape[1]
            if "nn" in ops:
                ms = timeit(lambda: pu.three_nn(unk, kn))
                print(f"three_nn n={n:6d} m={m:5d}: {ms:8.3f} ms")
            if "interp" in ops:
                C = fc[3 - lv]
                d, i = pu.three_nn(unk, kn)
                w = torch.rand(B, n, 3, device=dev)
                f = torch.randn(B, C, m, device=dev)
                ms = timeit(lambda: pu.three_interpolate(f, i, w))
                by = B * (24 * n + 4 * C * m + 4 * C * n)
                print(f"three_interpolate C={C:5d} n={n:6d}: {ms:8.3f} ms  {by / 1e9 / (ms / 1e3):8.1f} GB/s")




def model_breakdown(clouds="uniform", B=32, iters=5):
    """Per-call timing of one bench step (eager), every native call bracketed by HIP events."""
    import bench
    dev = torch.device("cuda:0")
    backbone, neck = bench.build_models(dev)
    _, points = bench.make_batch(B, 16384, clouds, 1234, dev)

    def step():
        bd = {'batch_size': B, 'points': points, 'points_per_sample_checked': True}
        return neck(backbone(bd))

    with torch.no_grad():
        step(); step()
        torch.cuda.synchronize()
        orig = _native.call
        recs = []

        def timed(name, stream, *args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(name, stream, *args)
            e1.record()
            recs.append((name, [a for a in args[:6] if isinstance(a, (int, float))], e0, e1))

        _native.call = timed
        try:
            for _ in range(iters):
                step()
        finally:
            _native.call = orig
        torch.cuda.synchronize()
    per = len(recs) // iters
    tot = 0.0
    for i in range(per):
        ms = sum(recs[i + k * per][2].elapsed_time(recs[i + k * per][3]) for k in range(iters)) / iters
        tot += ms
        print(f"{recs[i][0]:30s} {str(recs[i][1]):42s} {ms:8.3f} ms")
    print(f"sum of native calls: {tot:.3f} ms")


if __name__ == "__main__":
    main()
