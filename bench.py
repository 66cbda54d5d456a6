#!/usr/bin/env python3
"""bench.py — frames/s of PDM-SSD's point-cloud hot path on MI355X.

One "step" = one full PDM-SSD forward over one batch of synthetic KITTI-range clouds that are already
resident in HBM: PointNet2MSG backbone (4 SA-MSG + 4 FP layers: FPS, ball query, grouping, shared MLPs,
three-NN interpolation), the PDM neck (dilation, SH x Gaussian filling, scatter-add to the BEV grid,
normalise, height compression) and the hybrid head (BEV heat-map head on the neck's grid + point box head
on the backbone's point features, boxes decoded; NMS post-processing is not part of the step).  fp32,
inference (BN in eval mode), the configuration BASELINE.json's metric is quoted on (configs[2]): bs = 32
clouds of 16384 points per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU = pure data parallel over whole clouds (weak scaling, 32 clouds per rank, no data-path
collective; only the timing barriers).  Rank 0 prints ONE JSON line.  Started as plain `python bench.py
--gpus N` (no WORLD_SIZE in the environment) the process only acts as launcher: it starts N fresh child
processes, one rank per GPU, before anything touches the GPU (launch_ranks below), relays rank 0's line and
exits with the worst child code — the reference's equivalent is torch.distributed.launch feeding
tools/train.py:74-76 / pcdet/utils/common_utils.py:189-204.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from pdm_ssd_amd import _native, dist_utils, synthetic
from pdm_ssd_amd.detector_config import PDM_SSD_CFG, build_pdm_ssd
from pdm_ssd_amd.pipeline import PipelinedHotPath, overlapping_stream

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_GUIDE_COPY_GBS = 6290.0  # MI355X_MICROARCH.md: what a float4 device copy reaches — the achievable ceiling quoted beside the peak
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak (= fp32 vector peak)
VALU_F32_PEAK_TFLOPS = 157.3  # same figure for the packed-fp32 vector pipe (256 CUs x 256 flop/clk x 2.4 GHz)
NECK_CFG = PDM_SSD_CFG['MAP_TO_BEV']


def build_detector(device, seed=0):
    """PDM-SSD (pdm_ssd_amd/detector_config.py): PointNet2MSG -> PDM neck -> heat-map head + point box head."""
    torch.manual_seed(seed)
    model = build_pdm_ssd()
    with torch.no_grad():  # non-degenerate SH / scale head so the neck's arithmetic is fully exercised
        model.map_to_bev_module.coef.weight.normal_(0.0, 0.02)
    return model.to(device).eval()


def build_models(device, seed=0):
    """(backbone, neck) of the detector: the part of the step tests/test_modules_gpu.py checks at full size."""
    model = build_detector(device, seed)
    return model.backbone_3d, model.map_to_bev_module


def make_batch(B, N, kind, seed0, device):
    gen = synthetic.uniform_clouds if kind == "uniform" else synthetic.lidar_like_clouds
    clouds = gen(B, N, seed0)
    pts = torch.from_numpy(synthetic.to_batch_points(clouds)).to(device)
    return clouds, pts


# ----------------------------------------------------------------------------- per-op accounting

def algorithmic_bytes(name, a):
    """SURVEY.md section 8 D4 formulas; `a` = the integer/float arguments of the C-ABI call."""
    if name in ("pdm_ball_query", "pdm_ball_query_grid", "pdm_ball_query_grid_prebuilt"):
        b, n, m, _r, ns = a[:5]
        return b * (12 * n + 12 * m + 4 * m * ns)
    if name == "pdm_group_concat":
        b, n, m, c, ns = a[:5]
        # idx + source once (xyz and C feature rows) + centres + the (3+C) output channels
        return b * (4 * m * ns + 12 * n + 4 * c * n + 12 * m + 4 * (3 + c) * m * ns)
    if name == "pdm_group_points":
        b, c, n, m, ns = a[:5]
        return b * (4 * m * ns + 4 * c * n + 4 * c * m * ns)
    if name == "pdm_gather_points":
        b, c, n, m = a[:4]
        return b * (4 * m + 4 * c * n + 4 * c * m)
    if name in ("pdm_furthest_point_sampling", "pdm_furthest_point_sampling_ws"):
        b, n, m = a[:3]
        return b * (12 * n + 4 * m)
    if name in ("pdm_three_nn", "pdm_three_nn_grid", "pdm_three_nn_grid_prebuilt"):
        b, n, m = a[:3]
        return b * (12 * n + 12 * m + 24 * n)
    if name == "pdm_three_interpolate":
        b, c, m, n = a[:4]
        return b * (24 * n + 4 * c * m + 4 * c * n)
    if name in ("pdm_scatter_bev", "pdm_gather_bev"):
        # SURVEY D4 "PDM scatter": point inputs + the grid written once (the atomic read-modify-write traffic
        # 4*C*P*K of the scatter form is reported separately, see pdm_atomics_section)
        B, P, C, deg = a[:4]
        W, H, D = a[17:20]
        return B * (12 * P + 4 * C * P + 4 * (deg + 1) ** 2 * P + 4 * P) + 4 * B * C * D * H * W
    if name == "pdm_bev_depthwise3x3":
        B, H, W, C = a[:4]
        return 8 * B * H * W * C
    if name == "pdm_bev_head_fused":
        B, H, W, C = a[:4]
        return 4 * B * H * W * C
    if name == "pdm_bev_normalize":
        B, C, W, H, D = a[:5]
        return B * H * W * D * (8 * C + 4)
    return 0


class OpTimer:
    """Brackets every C-ABI call with HIP events on the stream the kernel is launched on."""

    def __init__(self):
        self.records = []
        self._orig = None

    def __enter__(self):
        self._orig = _native.call

        def timed(name, stream, *args):
            s = torch.cuda.current_stream()
            assert s.cuda_stream == stream or stream == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            self._orig(name, stream, *args)
            e1.record(s)
            ints = [x for x in args if isinstance(x, (int, float))]
            if name in ("pdm_rows_mlp_fused", "pdm_rows_mlp_fused_pair"):   # one entry point, several kernels: keep its shapes apart (fused.rows_tag)
                name = f"{name}[{args[3]} layers, {args[1]} in, {args[0]} rows]"
            self.records.append((name, ints, e0, e1))

        _native.call = timed
        return self

    def __exit__(self, *exc):
        _native.call = self._orig

    def summary(self, steps):
        torch.cuda.synchronize()
        agg = {}
        for name, ints, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            d = agg.setdefault(name, {"calls": 0, "ms": 0.0, "bytes": 0})
            d["calls"] += 1
            d["ms"] += ms
            d["bytes"] += algorithmic_bytes(name.split("[")[0], ints)
        out = []
        for name, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            ms_step = d["ms"] / steps
            out.append({"op": name, "calls_per_step": d["calls"] // steps, "ms_per_step": round(ms_step, 4),
                        "alg_MB_per_step": round(d["bytes"] / steps / 1e6, 3),
                        "GBps": round(d["bytes"] / steps / 1e9 / (ms_step / 1e3), 1) if ms_step > 0 else None})
        return out


def mlp_flops(seq):
    """2 * sum(cin * cout) of the 1x1 convs of a shared MLP = algorithmic FLOPs per position (unpadded)."""
    return 2 * sum(m.in_channels * m.out_channels for m in seq if isinstance(m, (torch.nn.Conv1d, torch.nn.Conv2d)))


def model_flops(backbone, B, N):
    """Algorithmic GFLOP per step of the SA and FP shared MLPs."""
    sa, fp, n = 0.0, 0.0, N
    level_n = [N]
    for m in backbone.SA_modules:
        for g, mlp in zip(m.groupers, m.mlps):
            sa += B * m.npoint * g.nsample * mlp_flops(mlp)
        level_n.append(m.npoint)
    for k, m in enumerate(backbone.FP_modules):
        fp += B * level_n[k] * mlp_flops(m.mlp)
    return sa / 1e9, fp / 1e9


def reference_op_section(backbone, points, B, iters=5):
    """The API-exact operators at the PointNet2MSG shapes, as the reference's QueryAndGroup issues them:
    per SA scale one ball_query and two grouping_operation calls (xyz^T and features).  This is the
    'ball_query + group_points vs HBM' figure of BASELINE.json, measured with HIP events per call."""
    from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
    xyz = points[:, 1:4].contiguous().view(B, -1, 3)
    feats = points[:, 4:].contiguous().view(B, -1, 1).permute(0, 2, 1).contiguous()
    chans = [1, 96, 256, 512]
    plan = []
    for k, m in enumerate(backbone.SA_modules):
        new_xyz = m.sample(xyz)
        f = feats if k == 0 else torch.randn(B, chans[k], xyz.shape[1], device=xyz.device)
        for g in m.groupers:
            plan.append((g.radius, g.nsample, xyz, new_xyz, f, xyz.transpose(1, 2).contiguous()))
        xyz = new_xyz
    from pdm_ssd_amd.pointnet2_batch import pointnet2_batch_hip as ext
    with OpTimer() as t:
        for it in range(iters + 1):
            if it == 1:
                t.records.clear()  # first pass = warm-up
            with pu.shared_search_grids():   # every pass is a new batch: one grid build per SA level, shared by its two radii
                for radius, ns, x, nx, f, xt in plan:
                    idx = pu.ball_query(radius, ns, x, nx)
                    pu.grouping_operation(xt, idx)
                    pu.grouping_operation(f, idx)
        ops = t.summary(iters)
    ms_events = sum(o["ms_per_step"] for o in ops)
    mb = sum(o["alg_MB_per_step"] for o in ops)
    gp = [o for o in ops if o["op"] == "pdm_group_points"][0]

    # The same 26 API calls (+ the 2 shared grid builds) as ONE hipGraph, replayed: the device time of the sequence with
    # nothing of the host between the launches (a per-call event pair costs 2-4 us of stream time per call, ~75 us over
    # the sequence, and Python needs ~14 us per call: the eager loop is host-bound).  This is the figure the fraction of
    # the HBM peak is taken from; the per-call event sums stay in `ops`.
    def sequence():
        with pu.shared_search_grids():
            for radius, ns, x, nx, f, xt in plan:
                idx = pu.ball_query(radius, ns, x, nx)
                pu.grouping_operation(xt, idx)
                pu.grouping_operation(f, idx)
    ms, how = ms_events, "sum of per-call HIP-event times (graph capture failed)"
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sequence()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            sequence()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms, how = e0.elapsed_time(e1) / reps, "hipGraph replay of the whole sequence (device time incl. the 27 launch boundaries)"
    except Exception as e:
        print(f"[bench] API-exact sequence: graph capture failed ({type(e).__name__}: {e})", file=sys.stderr)
    # The four SA levels are independent of each other once the sampled sets exist (the reference's coordinate chain): the same 26
    # calls as four chains — one per level: its grid build, two searches, four copies — on four streams inside one hipGraph, fork and
    # join only (no edge between chains).  Reported BESIDE the one-stream figure, not instead of it.
    conc = None
    try:
        cur = torch.cuda.current_stream()
        # which SA levels share a chain: "1|2|34" (default) = levels 1 and 2 on their own streams, 3 and 4 one after the other on a
        # third; "1|2|3|4" = one chain per level.  Measured with tools/diag/api_block.py --chains: 399 / 406 / 434 us for three / four /
        # two chains (uniform clouds) — a hipGraph runs at most four branches side by side on ROCm 7.2, and the latency-bound
        # launches of a chain wait for whole rounds of workgroups beside another chain's streaming copies, so more chains is not better.
        spec = os.environ.get("PDM_BENCH_CHAIN", "1|2|34")
        chains = [[int(c) - 1 for c in part] for part in spec.split("|")]
        assert sorted(l for ch in chains for l in ch) == list(range(len(plan) // 2)), spec
        streams = [torch.cuda.Stream() for _ in chains]
        def by_level():
            for levels, st in zip(chains, streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    with pu.shared_search_grids():
                        for li in levels:
                            for radius, ns, x, nx, f, xt in plan[2 * li:2 * li + 2]:
                                idx = pu.ball_query(radius, ns, x, nx)
                                pu.grouping_operation(xt, idx)
                                pu.grouping_operation(f, idx)
            for st in streams:
                cur.wait_stream(st)
        by_level()
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            cur = torch.cuda.current_stream()
            by_level()
        g2.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g2.replay()
        e1.record()
        torch.cuda.synchronize()
        cms = e0.elapsed_time(e1) / 20
        conc = {"ms_per_step": round(cms, 4), "GBps": round(mb / cms, 1), "frac_of_hbm_peak": round(mb / cms / HBM_PEAK_GBS, 4),
                "chains": spec,
                "timed_as": "hipGraph replay, the SA levels as independent chains on separate streams (fork / join only): " + spec}
    except Exception as e:
        print(f"[bench] API-exact sequence by level: graph capture failed ({type(e).__name__}: {e})", file=sys.stderr)
    # (Measured and dropped: the same 26 calls with the 8 searches on a second stream, every group_points call behind the event of
    #  its own ball_query, replayed as one graph: 0.503 ms against 0.481 on one stream — the cross-branch edges cost more than the
    #  overlap of the latency-bound searches with the copies returns.)
    # ball query is not bandwidth-bound as the reference states it: N * M distance evaluations of 8 flop per call
    # (SURVEY D3); the grid form visits only the cells a ball can reach, so its rate is quoted in reference-form
    # evaluations per second ("effective") next to the exhaustive scan's real rate
    evals = {"pdm_ball_query": 0.0, "pdm_ball_query_grid_prebuilt": 0.0}
    for radius, ns, x, nx, f, xt in plan:
        n, m = x.shape[1], nx.shape[1]
        evals["pdm_ball_query_grid_prebuilt" if 2048 <= n <= 131072 else "pdm_ball_query"] += float(B) * n * m
    bq = {}
    for o in ops:
        if o["op"] in evals and o["ms_per_step"] > 0:
            build_ms = sum(x["ms_per_step"] for x in ops if x["op"] == "pdm_grid_build") if "grid" in o["op"] else 0.0
            rate = evals[o["op"]] / ((o["ms_per_step"] + build_ms) * 1e-3)
            bq[o["op"]] = {"reference_form_evals_per_step": evals[o["op"]], "Gevals_per_s": round(rate / 1e9, 1),
                           # a fraction of the vector peak only where every evaluation is executed; the grid form skips most
                           "frac_of_valu_peak_at_8_flop_per_eval": None if "grid" in o["op"] else round(rate * 8 / 1e12 / VALU_F32_PEAK_TFLOPS, 4),
                           "kind": "effective (grid-pruned incl. its share of pdm_grid_build, same indices)" if "grid" in o["op"] else "executed (exhaustive scan)"}
    launches = sum(o["calls_per_step"] for o in ops)
    floor_ms = mb / 6200.0 + launches * 1.5e-3     # bytes at the 6.2 TB/s a streaming kernel reaches + 1.5 us per launch boundary
    return {"ops": ops, "ms_per_step": round(ms, 4), "timed_as": how, "ms_per_step_sum_of_event_pairs": round(ms_events, 4),
            "alg_MB_per_step": round(mb, 2),
            "GBps": round(mb / ms, 1), "frac_of_hbm_peak": round(mb / ms / HBM_PEAK_GBS, 4),
            "frac_of_hbm_peak_sum_of_event_pairs": round(mb / ms_events / HBM_PEAK_GBS, 4),
            "levels_concurrent": conc,
            "floor": {"ms": round(floor_ms, 4), "frac_of_hbm_peak": round(mb / floor_ms / HBM_PEAK_GBS, 4),
                      "how": f"{mb:.0f} MB / 6.2 TB/s (MI355X_MICROARCH.md: what plain 256-byte stores and a float4 copy reach) + "
                             f"{launches} launches x 1.5 us (dependent-launch boundary)"},
            "ball_query_distance_evals": bq,
            "note": "8 ball_query (+ 2 grid builds, one per level that uses the grid) + 16 group_points launches at bs=%d; "
                    "target >= 0.60" % B}, gp


# ----------------------------------------------------------------------------- CPU baseline

def cpu_baseline(model, N, kind, frames=2, threads=None):
    """Times the CPU statement of the same step on `frames` clouds: oracle C operators (OpenMP) for the backbone and
    neck (oracle/cpu_backbone.py), the detector's own torch layers on the CPU for the hybrid head."""
    import copy

    from oracle import cpu_backbone, cpu_oracle
    cpu_oracle.build()
    m = copy.deepcopy(model).cpu().eval()
    gen = synthetic.uniform_clouds if kind == "uniform" else synthetic.lidar_like_clouds
    all_threads = cpu_oracle.max_threads()
    usable = usable_cpus()
    if threads is None:
        # All-cores leg.  The oracle's FPS (the dominant operator on the CPU: 4095 dependent iterations per cloud) is parallel over
        # CLOUDS only, the other operators over clouds x centres: the sample holds one cloud per usable core (at most 64), so every
        # core has a cloud to sample; threads beyond the cores this process may use (cgroup quota / affinity) add nothing.
        threads = max(1, min(all_threads, usable))
        frames = max(frames, min(threads, 64))
    clouds = gen(frames, N, 4321)
    cpu_oracle.set_threads(threads)
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    with torch.no_grad():
        out = cpu_backbone.backbone_forward(m.backbone_3d, clouds)
        sf = cpu_backbone.neck_forward(m.map_to_bev_module, out['sa_xyz'], out['sa_features'])
        m.dense_head({'spatial_features': torch.from_numpy(sf)})
        coords = torch.from_numpy(synthetic.to_batch_points(clouds)[:, :4])
        m.point_head({'batch_size': frames, 'point_features': torch.from_numpy(out['point_features']), 'point_coords': coords})
    dt = time.perf_counter() - t0
    cpu_oracle.set_threads(all_threads)
    torch.set_num_threads(all_threads)
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "host": {"omp_max_threads": all_threads, "usable_cpus": usable, "os_cpu_count": os.cpu_count()},
            "fps_parallelism": min(frames, threads),
            "sample": f"{frames} cloud(s) x {N} pts, same step (oracle C operators with OpenMP + torch-CPU layers) on {threads} "
                      f"thread(s): FPS parallel over clouds ({min(frames, threads)}-way), the other operators over clouds x centres; "
                      f"{dt:.1f} s wall; the reference itself has no CPU path for these operators"}


def usable_cpus():
    """CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota when one is set (a container that sees
    128 CPUs but owns a 16-CPU share runs 128 OpenMP threads at the speed of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota != "max" and int(quota) > 0:
                n = min(n, max(1, int(quota) // int(period)))
            break
        except (OSError, ValueError):
            continue
    return n


# ----------------------------------------------------------------------------- training step (configs[3])

def synthetic_gt_boxes(B, M, seed, device):
    """(B, M, 8) [x, y, z, dx, dy, dz, heading, class] KITTI-like boxes for the head's target assignment."""
    rng = np.random.default_rng(seed)
    sizes = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], dtype=np.float32)
    cls = rng.integers(1, 4, (B, M))
    gt = np.zeros((B, M, 8), dtype=np.float32)
    gt[..., 0] = rng.uniform(5, 65, (B, M)); gt[..., 1] = rng.uniform(-35, 35, (B, M)); gt[..., 2] = rng.uniform(-1.6, -0.8, (B, M))
    gt[..., 3:6] = sizes[cls - 1] * rng.uniform(0.9, 1.1, (B, M, 3))
    gt[..., 6] = rng.uniform(-np.pi, np.pi, (B, M))
    gt[..., 7] = cls
    return torch.from_numpy(gt).to(device)


def train_bench(args, model, points, B, N, rank, world, local_rank, device, steps=None, warmup=None, scaling="weak"):
    """Training step (BASELINE configs[3]): the detector in train mode — autograd graph over the HIP operators (fused
    QueryAndGroup forward, inverted-index backward kernels, PDM scatter + its gather-form backward), torch modules
    for the MLPs under bf16 autocast, coordinates and indices in fp32, the hybrid head's losses (point focal +
    smooth-L1 over points_in_boxes targets, heat-map focal) — AdamW, one gradient all-reduce per step (DDP, single
    bucket) when world > 1."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    # --train-graph (one process, one GPU): the whole step (forward, losses, backward, AdamW) captured into a hipGraph after
    # the warm-up steps and REPLAYED; possible because the step has no host synchronisation and fixed shapes.  Measured:
    # 27.3 ms per step against 26.2 eager — hipGraphLaunch of the ~1500-node graph costs the host 22.6 ms (18.2 for the eager
    # Python loop) and the device-side launch-to-launch latency of the ~600 tiny kernels does not drop.  Not the default.
    want_graph = args.train_graph and world == 1 and not args.train_profile and not args.train_host_profile
    # torch's fused multi-tensor AdamW (one launch chain for all parameters; PDM_BENCH_ADAMW=foreach for the per-operation form:
    # same update, 21.8 instead of 21.7 ms per step and 1 ms more host time)
    opt = torch.optim.AdamW(params, lr=1e-3, capturable=want_graph, fused=os.environ.get("PDM_BENCH_ADAMW", "fused") == "fused" or None)
    # The reference's step fetches a NEW batch every iteration (tools/train_utils/train_utils.py:33) and clips the gradient norm
    # (:58-62, GRAD_NORM_CLIP = 10 in OpenPCDet's configs): NB distinct clouds + box sets rotate through static buffers (one
    # pdm_copy_many launch per step, as Bench._advance does for the inference step), so the cloud whose sampling chain runs on the
    # side stream really is the next batch, and every step ends with clip_grad_norm_ (foreach form: the norm stays on the device).
    NB = 4
    batches = [points] + [make_batch(B, N, args.clouds, 1234 + rank * B + 1000 * i, device)[1] for i in range(1, NB)]
    gts = [synthetic_gt_boxes(B, 12, 99 + rank + 17 * i, device) for i in range(NB)]
    cur_pts, next_pts, gt_boxes = batches[0].clone(), batches[1].clone(), gts[0].clone()
    backbone = model.backbone_3d
    grad_clip = float(os.environ.get("PDM_BENCH_GRAD_CLIP", "10"))

    def advance(i):
        """Static buffers <- batch i (consumed by this step) and batch i + 1 (sampled for the next step)."""
        _native.copy_many([cur_pts, gt_boxes, next_pts], [batches[i % NB], gts[i % NB], batches[(i + 1) % NB]])

    class Step(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, pts, sampled=None):
            bd = {'batch_size': B, 'points': pts, 'points_per_sample_checked': True, 'gt_boxes': gt_boxes}
            if sampled is not None:
                bd['sampled_xyz'] = sampled
            ret, tb, disp = self.m(bd)
            return ret['loss']

    stepper = Step(model)
    if world > 1:
        stepper = torch.nn.parallel.DistributedDataParallel(stepper, device_ids=[local_rank], bucket_cap_mb=64,
                                                            gradient_as_bucket_view=True)

    # The sampling chain (FPS + gather, one workgroup per cloud) needs no gradient: the chain of the NEXT batch
    # runs on a side stream under this batch's forward/backward.
    side = overlapping_stream(device)     # pdm_ssd_amd/pipeline.py: a stream on a hardware queue of its own
    state = {"sampled": None, "it": 0}

    def sample_next():
        with torch.no_grad():
            return backbone.sample_chain(next_pts[:, 1:4].contiguous().view(B, -1, 3))

    if not args.serial:
        next_pts.copy_(batches[0])       # the first step consumes batch 0: its chain is the "next" one now
        state["sampled"] = sample_next()
        torch.cuda.synchronize()

    def step_body():
        opt.zero_grad(set_to_none=True)
        if args.serial:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = stepper(cur_pts)
        else:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                nxt = sample_next()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = stepper(cur_pts, state["sampled"])
        loss.backward()
        if grad_clip > 0:
            torch.nn.utils.clip_grad_norm_(params, grad_clip, foreach=True)
        opt.step()
        if not args.serial:
            main.wait_stream(side)
            if state.get("static"):      # captured step: the next step reads the same buffers, refreshed in place
                for dst, src in zip(state["sampled"], nxt):
                    dst.copy_(src)
            else:
                for t in nxt:
                    t.record_stream(main)
                state["sampled"] = nxt
        return loss

    def step():
        advance(state["it"])
        state["it"] += 1
        return step_body()

    first_loss = None
    for _ in range(max(1, warmup)):
        l0 = step()
        first_loss = l0.detach().clone() if first_loss is None else first_loss      # (read back after the timed region)
    dist_utils.barrier()
    if args.train_profile:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        if os.environ.get("PDM_TRAIN_PROFILE_OPS") == "2":
            # every aten operator of ONE step (forward and backward on this thread) with the repo line that issued it:
            # which torch kernels are left in the step, and where they come from
            import collections, traceback
            from torch.utils._python_dispatch import TorchDispatchMode
            seen = collections.Counter()

            class Log(TorchDispatchMode):
                def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                    out = func(*args, **(kwargs or {}))
                    name = func.__name__ if hasattr(func, "__name__") else str(func)
                    t = next((a for a in args if isinstance(a, torch.Tensor)), None)
                    if t is not None and not t.is_cuda:
                        return out
                    frames = [fr for fr in traceback.extract_stack() if "/pdm_ssd_amd/" in fr.filename or fr.filename.endswith("bench.py")]
                    site = " <- ".join(f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in frames[::-1][:2]) if frames else "(autograd)"
                    shape = tuple(t.shape) if t is not None else ()
                    seen[(str(func), shape, str(t.dtype) if t is not None else "", site)] += 1
                    return out

            torch.autograd.set_multithreading_enabled(False)
            with Log():
                step()
            torch.cuda.synchronize()
            if rank == 0:
                with open(args.train_profile, "w") as f:
                    skip = ("aten.view", "aten.detach", "aten.as_strided", "aten.t.", "aten.transpose", "aten.permute", "aten.reshape", "aten._unsafe_view",
                            "aten.expand", "aten.slice.", "aten.select.", "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.empty", "aten.movedim",
                            "aten.unbind", "aten.split", "aten.narrow", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.numel", "aten.dim")
                    for (name, shape, dt, site), cnt in sorted(seen.items(), key=lambda kv: (-int(torch.Size(kv[0][1]).numel()), kv[0][0])):
                        if not any(name.startswith(p_) for p_ in skip):
                            f.write(f"{cnt:3d} x {name:42s} {str(shape):28s} {dt:16s} {site}\n")
            return
        by_op = os.environ.get("PDM_TRAIN_PROFILE_OPS") == "1"     # torch operators with their call sites instead of kernels
        acts = [ProfilerActivity.CPU, ProfilerActivity.CUDA] if by_op else [ProfilerActivity.CUDA]
        with profile(activities=acts, with_stack=by_op, record_shapes=by_op) as prof:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
        if rank == 0:
            with open(args.train_profile, "w") as f:
                f.write("# 3 steady-state train steps (divide totals by 3 for one step)\n")
                if by_op:
                    f.write(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=60,
                                                                        max_src_column_width=110))
                else:
                    f.write(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=110))
        return
    if args.train_host_profile:     # where the HOST spends a step (the step is within ~1 ms of host-issue-bound)
        import cProfile, pstats
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            step()
        pr.disable()
        torch.cuda.synchronize()
        # host synchronisations inside a step: torch's sync debug mode warns at every blocking call (a size read back for
        # a boolean-mask index, .item(), a pageable host->device copy ...); the innermost repo frame is listed
        import traceback, warnings
        syncs = []
        def note(message, category, filename, lineno, file=None, line=None):
            frames = [fr for fr in traceback.extract_stack() if "/pdm_ssd_amd/" in fr.filename or fr.filename.endswith("bench.py")]
            syncs.append(f"{message} <- " + " <- ".join(f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in frames[::-1][:3]))
        old_show = warnings.showwarning
        warnings.showwarning = note
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        try:
            step()
        finally:
            torch.cuda.set_sync_debug_mode("default")
            warnings.showwarning = old_show
        torch.cuda.synchronize()
        if rank == 0:
            with open(args.train_host_profile, "w") as f:
                f.write(f"# host synchronisations in one train step (forward + loss on this thread; the backward runs on the "
                        f"autograd thread): {len(syncs)}\n")
                for m in syncs:
                    f.write("#   " + m + "\n")
                f.write("# cProfile of 3 steady-state train steps (host side; divide by 3)\n")
                st = pstats.Stats(pr, stream=f)
                st.sort_stats("tottime").print_stats(45)
                st.sort_stats("cumulative").print_stats(70)
        return
    launch = "eager"
    graph = None
    if want_graph:
        try:
            torch.cuda.synchronize()
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            # A parameter's AccumulateGrad node keeps the stream it was created on, and it lives as long as any autograd graph
            # that uses the parameter: the heads keep their last predictions (forward_ret_dict), i.e. the graph of the last
            # eager step on the DEFAULT stream.  Drop those, so the step below makes the nodes anew on the capture stream.
            for m in model.modules():
                if isinstance(getattr(m, "forward_ret_dict", None), dict):
                    m.forward_ret_dict = {}
            import gc as _gc
            _gc.collect()
            with torch.cuda.stream(cap):
                state["static"] = True
                step()                     # one more eager step on the capture stream (allocator warm-up on that stream)
                cap.synchronize()
                graph = torch.cuda.CUDAGraph()
                opt.zero_grad(set_to_none=True)
                with torch.cuda.graph(graph, stream=cap):
                    static_loss = step_body()     # the input rotation (advance) stays outside the captured region
            torch.cuda.current_stream().wait_stream(cap)
            torch.cuda.synchronize()
            graph.replay(); graph.replay()
            torch.cuda.synchronize()
            launch = "hipGraph replay of the whole step (forward, losses, backward, AdamW)"
        except Exception as e:   # say so and time the eager loop instead
            import traceback
            print(f"[bench] train-step capture failed ({type(e).__name__}: {str(e).splitlines()[0]}); timing the eager loop\n"
                  + "".join(traceback.format_exc().splitlines(True)[-14:]), file=sys.stderr)
            graph = None
            state["static"] = False
            torch.cuda.synchronize()
    if graph is not None:
        def step():   # noqa: F811 - the timed loop below replays
            advance(state["it"])
            state["it"] += 1
            graph.replay()
            return static_loss
    import gc
    gc.collect()
    gc.disable()     # ~1500 launches per step are issued from Python: a generational collection inside the timed region is host time
    try:
        dist_utils.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        host_ms = (time.perf_counter() - t0) / steps * 1e3    # all launches issued; the device may still be working
        dist_utils.barrier()
        elapsed = dist_utils.max_over_ranks(time.perf_counter() - t0, device)
    finally:
        gc.enable()
    return {
        "metric": f"train frames/sec ({N}-pt clouds, bs={B}/GPU, bf16 autocast)", "value": round(world * B * steps / elapsed, 2),
        "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3), "host_issue_ms_per_step": round(host_ms, 3),
        "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "bf16 (MLPs) / f32 (coordinates, operators)", "data": "synthetic",
        "config": {"workload": f"configs[3]: PDM-SSD train step (PointNet2MSG + PDM neck + hybrid head losses), bs={B}/GPU x {N} "
                               f"pts, 12 synthetic boxes per cloud, {NB} distinct batches + box sets in rotation, clip_grad_norm_({grad_clip:g}), "
                               "AdamW (torch fused multi-tensor), DDP gradient all-reduce over RCCL", "parallelism": f"dp{world}",
                   "global_batch": world * B, "launch": launch,
                   "overlap": "none" if args.serial else "FPS chain of the next batch on a side stream"},
        # the loss of the first warm-up step beside the last timed one: a step that trains on stale weights (a cache of packed weights
        # that misses the optimizer's update did exactly that for a while) shows here as a loss that does not fall
        "first_loss": float(first_loss), "final_loss": float(loss.detach())}


# ----------------------------------------------------------------------------- launcher

def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N child processes of this same script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (one process per GPU, rendezvous on 127.0.0.1),
    wait for them and return the worst exit code.  The parent never initialises the GPU and never execs: the
    children are ordinary fresh processes.  Rank 0's stdout is inherited, so its JSON line is this command's line;
    the other ranks' stdout goes to stderr.  If a rank dies, the others are terminated (by pid)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    worst, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 128 - rc)
                print(f"[bench] rank {r} exited with code {rc}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    return worst


def rendezvous_only(args):
    """--rendezvous-only: the launch + timing protocol of the benchmark without a kernel (gloo when there is no GPU):
    process group from the environment, barrier, MAX-over-ranks, ONE line from rank 0.  Used by tests/test_dp_gloo.py
    to drive `python bench.py --gpus 2` on a CPU-only box, and as a launch diagnostic on a GPU node."""
    rank, world, local_rank = dist_utils.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: start with `python bench.py --gpus N` or torchrun"
    dist_utils.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))
    dist_utils.barrier()
    elapsed = dist_utils.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "rendezvous only (no kernels)", "n_gpus": world, "steps": args.steps,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "backend": dist.get_backend() if dist.is_initialized() else None}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------- the timed step

class Bench:
    """One configuration (B clouds of N points, cloud kind) set up for timing: distinct synthetic batches rotated
    through static input buffers (every pipeline stage sees a DIFFERENT batch every step), the pipelined step
    captured in a hipGraph."""

    NBATCH = 4

    def __init__(self, model, B, N, kind, depth, device, seed0, serial=False, graph=True, autotune=True):
        self.model, self.B, self.N, self.kind, self.device = model, B, N, kind, device
        self.backbone, self.neck = model.backbone_3d, model.map_to_bev_module
        self.serial, self.mode = serial, "eager"
        self.batches = [make_batch(B, N, kind, seed0 + 1000 * i, device)[1] for i in range(self.NBATCH)]
        fps_wgs = (depth - 1) * B * ((N + 16383) // 16384)
        if depth >= 3 and not (1024 < N <= 131072 and (N <= 16384 or fps_wgs <= _native.lib().pdm_fps_max_coresident_workgroups())):
            print(f"[bench] resumable FPS segments need 1024 < points <= 131072 and co-resident workgroups; "
                  f"{N} points x {B} clouds -> --pipeline-depth 2", file=sys.stderr)
            depth = 2
        self.depth = depth
        self.pipe = PipelinedHotPath(self.backbone, self.neck, depth=depth, dense_head=model.dense_head, point_head=model.point_head)
        self.nstage = depth + 1 if depth >= 3 else 3           # input buffers: current batch + the ones in flight
        self.inputs = [self.batches[k % self.NBATCH].clone() for k in range(self.nstage)]
        self.step_no = 0
        self.hoisting = None
        with torch.no_grad():
            counts = torch.bincount(self.batches[0][:, 0].long(), minlength=B)   # the backbone's per-sample point-count
            assert int(counts.min()) == int(counts.max()) == N                    # check (host sync), once, untimed
            if autotune:
                self.hoisting = self.backbone.autotune_hoisting(self.batches[0], B)
            if not serial:
                if depth >= 3:
                    self.pipe.prime_segmented(self.inputs[:depth], B)
                else:
                    self.pipe.prime(self.inputs[0], B, points_next=self.inputs[1])
            self._run = self.step
            if graph:
                self._capture()

    def step_serial(self, points=None):
        bd = {'batch_size': self.B, 'points': self.inputs[0] if points is None else points, 'points_per_sample_checked': True}
        if self.model.point_head.wants_deferred_fp():    # as the detector's forward and the pipeline do: last FP module inside the head's launch
            bd['defer_last_fp'] = True
        bd = self.model.point_head(self.model.dense_head(self.neck(self.backbone(bd))))
        return bd['spatial_features'], bd['point_features'], bd['batch_box_preds'], bd['bev_heatmap']

    def step(self):
        if self.serial:
            return self.step_serial()
        i = self.inputs
        bd = self.pipe.step(i[0], i[1], self.B, extra={'points_per_sample_checked': True}, points_next2=i[2],
                            points_ahead=i[1:self.depth + 1] if self.depth >= 3 else None)
        return bd['spatial_features'], bd['point_features'], bd['batch_box_preds'], bd['bev_heatmap']

    def _advance(self):
        """The input buffers move one batch on: stage k now holds the batch that was at stage k + 1 (what the pipeline's
        state expects after a step), the last stage a new one.  One 10 MB device copy per stage, part of the step."""
        _native.copy_many(list(self.inputs), [self.batches[(self.step_no + k) % self.NBATCH] for k in range(len(self.inputs))])   # one launch
        self.step_no += 1

    def _capture(self):
        try:
            for _ in range(2):
                self._advance()
                self.step()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._advance()
                self.step()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):     # recorded, not executed: the pipeline's state does not move
                self.static_out = self.step()
            self._run, self.mode = self.graph.replay, "hipGraph"
            self.run()
            torch.cuda.synchronize()
        except Exception as e:  # capture unsupported -> measure eager, say so
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            self._run, self.mode = self.step, "eager"
            torch.cuda.synchronize()

    def run(self):
        """One step: rotate the inputs (the pipeline's stages hold batches i, i+1, ...), then the step."""
        self._advance()
        out = self._run()
        if out is not None:       # eager launch: fresh tensors every step (a graph replay writes self.static_out)
            self.last_out = out

    OUTPUTS = ("spatial_features", "point_features", "batch_box_preds", "bev_heatmap")

    def outputs(self):
        """The four result tensors of the LAST run(): the captured step's static outputs under hipGraph replay."""
        return self.static_out if self.mode == "hipGraph" else self.last_out

    def consumed_batch(self):
        """The batch the last run() processed (what _advance() put into stage 0)."""
        return self.batches[(self.step_no - 1) % self.NBATCH]

    def verify(self, steps=2):
        """Untimed self-check of the object that is timed: `steps` more run() calls (input rotation + the pipelined,
        possibly graph-replayed step); after each, the step's outputs must equal BIT FOR BIT what the plain serial
        detector loop returns for the batch that step consumed (backbone -> neck -> dense head -> point head on the
        current stream: the reference's pcdet/models/detectors/point_rcnn.py:9-11).  Same kernels either way, so any
        difference is a stale buffer or a missing stream edge.  Returns the `verified` block of the bench line or
        raises AssertionError."""
        with torch.no_grad():
            for s in range(steps):
                self.run()
                torch.cuda.synchronize()
                got = [t.clone() for t in self.outputs()]
                want = self.step_serial(self.consumed_batch())
                torch.cuda.synchronize()
                for name, g, w in zip(self.OUTPUTS, got, want):
                    assert g.data_ptr() != w.data_ptr(), f"{name}: serial result aliases the step's output"
                    assert g.shape == w.shape, f"{name}: shape {tuple(g.shape)} vs serial {tuple(w.shape)}"
                    if not torch.equal(g, w):
                        bad = int((g != w).sum())
                        raise AssertionError(f"step {self.step_no} ({self.mode}, depth {self.depth}): {name} differs from the "
                                             f"serial detector in {bad} of {g.numel()} elements "
                                             f"(max |diff| {float((g.float() - w.float()).abs().max()):.3e})")
        return {"vs_serial": "bit-equal", "steps": steps, "outputs": list(self.OUTPUTS),
                "launch": self.mode, "pipeline_depth": 1 if self.serial else self.depth}

    def timed(self, steps, warmup, barrier=lambda: torch.cuda.synchronize()):
        with torch.no_grad():
            for _ in range(warmup):
                self.run()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.run()
            barrier()
            return time.perf_counter() - t0

    def median_ms(self, iters=50):
        """Median over `iters` individually timed steps (HIP events on the launch stream), SURVEY D2."""
        with torch.no_grad():
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
            for e0, e1 in ev:
                e0.record()
                self.run()
                e1.record()
            torch.cuda.synchronize()
        t = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
        return t[len(t) // 2], t[0], t[-1]


def measured_copy_bandwidth(device, mib=1024, iters=10):
    """Device-to-device float4 copy of `mib` MiB through this library's own copy kernel: the HBM figure the box
    sustains (read + write bytes per second), reported beside the nominal 8 TB/s."""
    n = mib * 1024 * 1024 // 4
    src, dst = torch.empty(n, dtype=torch.float32, device=device).normal_(), torch.empty(n, dtype=torch.float32, device=device)
    _native.copy_many([dst], [src])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _native.copy_many([dst], [src])
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 4 * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9


def pdm_atomics_section(model, bd_source, reps=5):
    """The PDM neck's scatter form (atomics-on-HBM, pdm_scatter_bev: memset + atomic adds + normalise — what training
    runs and what north_star names) timed on the sampled set of the current batch, beside the gather form the
    inference step uses.  Bytes: SURVEY D4's PDM formula, the atomic read-modify-write volume 4*C*P*K separately."""
    neck = model.map_to_bev_module
    with torch.no_grad():
        bd = {'sa_xyz': bd_source['sa_xyz'], 'sa_features': bd_source['sa_features']}
        out = {}
        for form in ("gather", "scatter"):
            neck.use_gather = form == "gather"
            with OpTimer() as t:
                for it in range(reps + 1):
                    if it == 1:
                        t.records.clear()
                    neck(dict(bd))
                ops = [o for o in t.summary(reps) if o["op"].startswith(("pdm_scatter", "pdm_gather", "pdm_bev_norm"))]
            out[form] = ops
        neck.use_gather = True
    xyz = bd_source['sa_xyz'][neck.source_layer]
    B, P = xyz.shape[0], xyz.shape[1]
    K = neck.dilation[0] * neck.dilation[1] * neck.dilation[2]
    rmw = 4.0 * neck.feature_dim * P * K * B
    sc = [o for o in out["scatter"] if o["op"] == "pdm_scatter_bev"]
    res = {"gather_form_ops": out["gather"], "scatter_form_ops": out["scatter"],
           "atomic_rmw_MB_per_step": round(rmw / 1e6, 1)}
    if sc and sc[0]["ms_per_step"] > 0:
        res["atomics_kernel"] = {"kernel": "pdm::pdm_scatter_kernel", "ms": sc[0]["ms_per_step"],
                                 "atomic_add_GBps": round(rmw / 1e9 / (sc[0]["ms_per_step"] * 1e-3), 1),
                                 "note": "memory-side fp32 atomic adds, 4 bytes of payload each"}
    return res


# ----------------------------------------------------------------------------- main

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling (SURVEY D2): a fixed global batch sharded over the ranks, G / N whole clouds per GPU "
                         "(the reference's DistributedSampler over a fixed dataset, pcdet/datasets/__init__.py:69-74); "
                         "0 = weak scaling with --batch clouds on every GPU")
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--clouds", choices=["uniform", "lidar"], default="uniform")
    ap.add_argument("--pipeline-depth", type=int, default=4, choices=[1, 2, 3, 4, 5],
                    help="batches whose sampling chain is in flight beside the feature path; >= 3 also cuts the level-1 "
                         "FPS into depth - 1 resumable segments run side by side (pdm_ssd_amd/pipeline.py)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-autotune", action="store_true",
                    help="keep the first SA layer of every level hoisted instead of choosing per level from the "
                         "neighbour density of the batch (PointNet2MSG.autotune_hoisting)")
    ap.add_argument("--serial", action="store_true", help="no cross-batch overlap of the FPS chain")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 4 instead: bf16-autocast forward+backward+AdamW step of the detector, "
                         "DistributedDataParallel gradient all-reduce over RCCL when --gpus > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-graph", action="store_true",
                    help="with --train at one GPU: capture the whole step into a hipGraph and time replays (measured slower than the eager loop)")
    ap.add_argument("--train-host-profile", metavar="FILE", default=None,
                    help="with --train: write a cProfile table of three steady-state steps (host side) to FILE and exit")
    ap.add_argument("--train-profile", metavar="FILE", default=None,
                    help="with --train: after the warm-up, run 3 steps under torch.profiler and write the per-kernel table "
                         "(steady state: MIOpen's one-off solver search stays outside) to FILE instead of timing")
    ap.add_argument("--cpu-frames", type=int, default=8, help="clouds of the bounded CPU-baseline sample (about 10 s of CPU work per leg)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra measurements of the default line (lidar-like clouds, configs[4] stress shape, "
                         "single-thread CPU leg, median over 50 steps)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch + barrier + max-over-ranks protocol only, no kernels (launch diagnostic / CPU test)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher BEFORE any torch.cuda / HIP call in this process
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.rendezvous_only:
        return rendezvous_only(args)

    if not torch.cuda.is_available():
        sys.exit(f"[bench] rank {os.environ.get('RANK', '0')}: no GPU visible (torch.cuda.is_available() is False); "
                 "the hot path has no CPU fallback")
    rank, world, local_rank = dist_utils.init_from_env(backend="nccl")  # 'nccl' is RCCL on ROCm
    if world != args.gpus:
        sys.exit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: start with `python bench.py --gpus {args.gpus}` "
                 f"(self-launching) or torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    _native.lib()  # fail loudly now if the HIP library is missing
    if os.environ.get("PDM_FP_CHAIN_MASK"):      # A/B knob: which FP shapes take the register-resident chain kernel
        _native.lib().pdm_tune_fp_chain_mask(int(os.environ["PDM_FP_CHAIN_MASK"]))

    if os.environ.get("PDM_DW_WG_PER_CU"):       # A/B knob: grid cap of the heat-map head's one-kernel form
        _native.lib().pdm_tune_rows_chain_dw_wg_per_cu(int(os.environ["PDM_DW_WG_PER_CU"]))

    if os.environ.get("PDM_RC_WG_PER_CU"):       # A/B knob: grid cap of the point head's chain kernels (workgroups per CU over the launch)
        _native.lib().pdm_tune_rows_chain_wg_per_cu(int(os.environ["PDM_RC_WG_PER_CU"]))

    B, N = args.batch, args.points
    scaling = "weak"
    if args.global_batch:
        if args.global_batch % world:
            sys.exit(f"[bench] --global-batch {args.global_batch} does not divide over {world} ranks (whole clouds per rank)")
        B, scaling = args.global_batch // world, "strong"
    model = build_detector(device)
    if os.environ.get("PDM_HM_FUSED_LOSS"):      # A/B knob: 0 = the heat-map head's targets and loss as torch kernels
        model.dense_head.use_fused_loss = os.environ["PDM_HM_FUSED_LOSS"] != "0"
    if os.environ.get("PDM_FP_HEAD_FUSION"):     # A/B knob: 0 = the last FP module and the point head as separate launches
        model.point_head.use_fp_fusion = os.environ["PDM_FP_HEAD_FUSION"] != "0"
    if args.train:
        _, points = make_batch(B, N, args.clouds, 1234 + rank * B, device)
        line = train_bench(args, model, points, B, N, rank, world, local_rank, device, scaling=scaling)
        if rank == 0 and line is not None:
            print(json.dumps(line))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    backbone, neck = model.backbone_3d, model.map_to_bev_module

    bench = Bench(model, B, N, args.clouds, args.pipeline_depth, device, seed0=1234 + rank * B, serial=args.serial,
                  graph=not args.no_graph, autotune=not args.no_autotune)
    elapsed = bench.timed(args.steps, args.warmup, barrier=dist_utils.barrier)
    elapsed = dist_utils.max_over_ranks(elapsed, device)
    LAUNCH_MODE[0] = bench.mode
    LAUNCH_MODE[1] = None if bench.hoisting is None else [d["use_pre"] for d in bench.hoisting]
    if not args.serial:
        bench.pipe.check_sampling()   # N > 16384: no cooperating FPS workgroup gave up waiting for a peer
    # untimed: the timed object (rotation + pipelined step under graph replay) against the serial detector, on every rank
    try:
        verified = bench.verify(2)
    except AssertionError as e:
        print(f"[bench] rank {rank}: VERIFICATION FAILED: {e}", file=sys.stderr, flush=True)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(3)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    frames_per_s = world * B * args.steps / elapsed
    extras = {}
    if not args.no_extras:
        med, lo, hi = bench.median_ms(max(50, args.steps))
        extras["ms_per_step_median_of_50"] = {"median": round(med, 4), "min": round(lo, 4), "max": round(hi, 4),
                                              "frames_per_s_at_median": round(B / med * 1e3, 1)}

    # in-situ pass: the timed (pipelined) step launched eagerly, every C-ABI call bracketed by HIP events on the stream it
    # is launched on — launch durations as they are inside the step, beside the other branches.  This pass decides which
    # entry point has the most device time in the step (`roofline`).
    insitu = []
    from pdm_ssd_amd import fused as _fused
    if not args.serial:
        with torch.no_grad():
            psteps = max(3, min(args.steps, 10))
            bench._advance(); bench.step(); torch.cuda.synchronize()
            with OpTimer() as timer:   # (no flop accounting here: it would put host synchronisations into the step)
                for _ in range(psteps):
                    bench._advance()
                    bench.step()
                insitu = timer.summary(psteps)

    # per-kernel pass (eager, serial, every C-ABI call bracketed by HIP events on its launch stream)
    points = bench.batches[0]
    with torch.no_grad():
        psteps = max(3, min(args.steps, 10))
        _fused.FLOP_COUNTER = {}
        with OpTimer() as timer:
            for _ in range(psteps):
                bench.step_serial(points)
            ops = timer.summary(psteps)
        executed = {k: v / psteps for k, v in _fused.FLOP_COUNTER.items()}
        insitu_flops = executed if insitu else {}   # executed flops per step are the serial pass's (same work per step)
        for o in insitu:
            if o["op"] in insitu_flops:
                o["executed_GFLOP_per_step"] = round(insitu_flops[o["op"]] / 1e9, 2)
                o["TFLOPs"] = round(insitu_flops[o["op"]] / 1e9 / o["ms_per_step"], 1)
        _fused.FLOP_COUNTER = None
        t0s = time.perf_counter()
        for _ in range(psteps):
            bench.step_serial(points)
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t0s) / psteps * 1e3
        bd_src = neck(backbone({'batch_size': B, 'points': points, 'points_per_sample_checked': True}))
        pdm = pdm_atomics_section(model, bd_src)
        copy_gbs = measured_copy_bandwidth(device)

    # HBM traffic per launch from the committed PMC passes (rocprofv3 cannot run inside this process): only a file
    # recorded for THIS shape is attached (tools/pmc_traffic.py stores batch / points / clouds)
    pmc, pmc_src = {}, None
    try:
        import glob as _glob
        for path in sorted(_glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
            doc = json.load(open(path))
            if doc.get("shape") == {"batch": B, "points": N, "clouds": args.clouds}:
                pmc, pmc_src = doc["kernels"], os.path.relpath(path, ROOT)
                break
    except Exception:
        pass

    def pmc_rows(name):
        """PMC rows of one kernel: a full instantiation ('pdm::rows_chain_kernel<8, 16, 16, 1, false>') matches only itself,
        a bare template / function name matches every instantiation of THAT name (not names it is a prefix of)."""
        return [v for k, v in pmc.items() if v["launches"] > 0 and (k == name or ("<" not in name and k.split("<")[0] == name))]

    def pmc_traffic(alternatives, together=()):
        """HBM bytes per CALL of an entry point.  `alternatives`: kernels of which each call launches ONE (the mean over
        their launches); `together`: kernels every call launches in addition (each one's per-launch mean is added)."""
        rows = [v for name in alternatives for v in pmc_rows(name)]
        if not rows:
            return None
        total = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in rows) / sum(v["launches"] for v in rows)
        for name in together:
            extra = pmc_rows(name)
            if extra:
                total += sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in extra) / sum(v["launches"] for v in extra)
        return int(total)

    # FLOPs: `executed` = what each entry point actually contracts (2 * real cin * cout per position and layer).
    sa_gf, fp_gf = model_flops(backbone, B, N)
    head_gf = B * N * (mlp_flops_linear(model.point_head.cls_layers) + mlp_flops_linear(model.point_head.box_layers)) / 1e9
    for o in ops:
        if o["op"] in executed:
            o["executed_GFLOP_per_step"] = round(executed[o["op"]] / 1e9, 2)
            o["TFLOPs"] = round(executed[o["op"]] / 1e9 / o["ms_per_step"], 1)
    mlp_ops = [o for o in ops if o["op"] in executed]
    flop_summary = {"reference_form_GFLOP_per_step": round(sa_gf + fp_gf + head_gf, 2),
                    "of_which": {"SA": round(sa_gf, 2), "FP": round(fp_gf, 2), "point_head": round(head_gf, 2)},
                    "executed_GFLOP_per_step": round(sum(executed.values()) / 1e9, 2),
                    "mlp_kernels_ms_per_step": round(sum(o["ms_per_step"] for o in mlp_ops), 4),
                    "note": "SA kernels run over compacted neighbour lists (ball_query's padding copies of the first hit are "
                            "not computed: max-pool over a multiset = over the set; bit-identical; uniform clouds hold one point "
                            "per ball, lidar-like ones 1.2-5.6); first-layer hoisting where it pays: W1 [f_nb ; dx] = (W1f f)[nb] "
                            "+ W1x dx (SA, chosen per level by autotune_hoisting), W1 interp(f) = interp(W1 f) (FP)"}
    KERNELS_OF = {"pdm_sa_mlp_fused": ("pdm::sa_mlp_fused_kernel", "pdm::sa_reg_mlp_kernel"),
                  "pdm_sa_mlp_fused_pre": ("pdm::sa_mlp_fused_kernel",),
                  "pdm_sa_mlp_packed": ("pdm::sa_packed_fused_kernel", "pdm::sa_reg_packed_kernel"),
                  "pdm_sa_mlp_packed_pair": ("pdm::sa_packed_pair_kernel", "pdm::sa_reg_packed_kernel"),   # both scales of a level
                  "pdm_fp_mlp_fused": ("pdm::fp_mlp_fused_kernel",),
                  "pdm_fp_mlp_fused_pre": ("pdm::fp_chain_kernel", "pdm::fp_mlp_fused_kernel"),   # FP1-2 chain, FP3-4 tiled
                  "pdm_bev_head_fused": ("pdm::rows_chain_kernel<8, 4, 4, 1, true>",),
                  "pdm_rows_mlp_fused_pair": ("pdm::rows_chain_pair_kernel",),     # the point head's two stacks in one launch
                  "pdm_rows_mlp_fused": ("pdm::rows_chain_kernel", "pdm::fp_mlp_fused_kernel", "pdm::rows_gemm_kernel<false>")}

    def kernels_of(op):
        """Kernel names behind an entry point; pdm_rows_mlp_fused picks by shape (fused_mlp.hip::pdm_rows_mlp_fused):
        3-layer chains over many rows -> rows_chain.hip, one wide layer -> rows_gemm.hip, else the general chain kernel."""
        base = op.split("[")[0]
        if base == "pdm_rows_mlp_fused" and "[" in op:
            layers = int(op.split("[")[1].split(" ")[0])
            rows = int(op.split(" in, ")[1].split(" ")[0])
            if layers == 3 and rows >= 8192:
                # the chain's instantiation carries the widths / 16: only the ones with this input width, no depthwise prologue
                cin = int(op.split(" layers, ")[1].split(" ")[0])
                inst = tuple(k for k in pmc if k.startswith(f"pdm::rows_chain_kernel<{cin // 16}, ") and k.endswith(", false>"))
                return inst or ("pdm::rows_chain_kernel",)
            return ("pdm::rows_gemm_kernel<false>",) if layers == 1 and rows >= 16384 else ("pdm::fp_mlp_fused_kernel",)
        return KERNELS_OF.get(base, ())

    def mfma_block(o):
        calls = o["calls_per_step"]
        per_launch_flop = executed[o["op"]] / calls
        per_launch_s = o["ms_per_step"] / 1e3 / calls
        ach = per_launch_flop / per_launch_s / 1e12
        kernels = kernels_of(o["op"])
        return {"bound": "mfma", "kernel": " + ".join(kernels) + f" (all launches of {o['op']})",
                "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": pmc_traffic(kernels) if kernels else None,
                "traffic_unit": "HBM bytes per launch (PMC)", "launches_per_step": calls,
                "avg_launch_us": round(per_launch_s * 1e6, 2), "alg_flop_per_launch": int(per_launch_flop),
                "ms_per_step": o["ms_per_step"], "flops_counted": "executed (hoisted first layer), unpadded"}

    def fps_block():
        """FPS: a dependent chain of m - 1 iterations per cloud, each N distance evaluations of 8 flop; bound by the
        latency of an iteration (VALU + cross-lane reductions + one barrier), quoted against the fp32 vector peak."""
        fo = [o for o in ops if o["op"] == "pdm_furthest_point_sampling"]
        if not fo:
            return None
        lv = [(m.npoint, n_in) for m, n_in in zip(backbone.SA_modules, [N] + [m.npoint for m in backbone.SA_modules][:-1])]
        flop = sum(8.0 * B * (m - 1) * n for m, n in lv)
        iters = sum(m - 1 for m, _ in lv)
        ms = fo[0]["ms_per_step"]
        ach = flop / (ms * 1e-3) / 1e12
        return {"bound": "valu/latency", "kernel": "pdm::fps_pruned_kernel + pdm::fps_reg_kernel (all launches of "
                "pdm_furthest_point_sampling: the four levels' chains, whole calls; the pipelined step runs level 1 as "
                "resumable jobs of the same kernel)",
                "achieved": round(ach, 3), "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / VALU_F32_PEAK_TFLOPS, 4),
                "traffic": pmc_traffic(("pdm::fps_pruned_kernel", "pdm::fps_reg_kernel")), "traffic_unit": "HBM bytes per launch (PMC)",
                "launches_per_step": fo[0]["calls_per_step"], "avg_launch_us": round(ms * 1e3 / fo[0]["calls_per_step"], 2),
                "alg_flop_per_launch": int(flop / fo[0]["calls_per_step"]), "ms_per_step": ms,
                "serial_iterations_per_step": iters, "us_per_iteration": round(ms * 1e3 / iters, 3),
                "distance_evals_per_s_G": round(flop / 8 / (ms * 1e-3) / 1e9, 1),
                "flops_counted": "reference form: every point against every new sample (the kernel skips groups whose "
                                 "bound proves nothing changes — exact — so fewer are executed)"}

    mfma_blocks = {o["op"]: mfma_block(o) for o in mlp_ops}
    roofline_mfma = max(mfma_blocks.values(), key=lambda b: b["ms_per_step"]) if mfma_blocks else None
    roofline_fps = fps_block()
    # `roofline` = the entry point with the most device time in the TIMED (pipelined) step, from the in-situ pass; its
    # block is rebuilt with the in-situ launch durations.  (serial run: the serial pass is the timed step)
    top = max(insitu, key=lambda o: o["ms_per_step"]) if insitu else None
    roofline = None
    if top is not None and top["op"] in insitu_flops:
        saved, executed = executed, insitu_flops
        roofline = mfma_block(top)
        executed = saved
        roofline["measured"] = "in situ: HIP events on the launch stream inside the pipelined step (eager launch of the timed step)"
    elif top is not None and top["op"].startswith("pdm_furthest_point_sampling"):
        lv = [(m.npoint, n_in) for m, n_in in zip(backbone.SA_modules, [N] + [m.npoint for m in backbone.SA_modules][:-1])]
        mine = [o for o in insitu if o["op"] == top["op"]][0]
        # the jobs form runs every iteration of the level-1 chain once per step (as depth - 1 segments of different batches)
        flop = 8.0 * B * (lv[0][0] - 1) * lv[0][1] if top["op"].endswith("_jobs") else sum(8.0 * B * (m - 1) * n for m, n in lv[1:])
        ach = flop / (mine["ms_per_step"] * 1e-3) / 1e12
        roofline = {"bound": "valu/latency", "kernel": f"pdm::fps_pruned_kernel (all launches of {top['op']})", "achieved": round(ach, 3),
                    "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / VALU_F32_PEAK_TFLOPS, 4),
                    "traffic": pmc_traffic(("pdm::fps_pruned_kernel",)), "launches_per_step": mine["calls_per_step"],
                    "avg_launch_us": round(mine["ms_per_step"] * 1e3 / mine["calls_per_step"], 2),
                    "alg_flop_per_launch": int(flop / mine["calls_per_step"]), "ms_per_step": mine["ms_per_step"],
                    "measured": "in situ: HIP events on the launch stream inside the pipelined step"}
    if roofline is None:
        cands = [b for b in (roofline_mfma, roofline_fps) if b is not None]
        roofline = max(cands, key=lambda b: b["ms_per_step"]) if cands else None

    with torch.no_grad():
        refops, gp = reference_op_section(backbone, points, B)
    gp_launch_bytes = gp["alg_MB_per_step"] * 1e6 / gp["calls_per_step"]
    gp_launch_s = gp["ms_per_step"] / 1e3 / gp["calls_per_step"]
    gp_gbs = gp_launch_bytes / gp_launch_s / 1e9
    roofline_hbm = {"bound": "hbm", "kernel": "pdm::group_points_rows_kernel + pdm::group_points_lds_kernel + pdm::group_points_v4_kernel "
                    "(all launches of pdm_group_points, the API-exact operator)",
                    "achieved": round(gp_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gp_gbs / HBM_PEAK_GBS, 4),
                    "frac_of_measured_copy": round(gp_gbs / copy_gbs, 4), "frac_of_guide_copy": round(gp_gbs / HBM_GUIDE_COPY_GBS, 4),
                    "traffic": pmc_traffic(("pdm::group_points_v4_kernel", "pdm::group_points_lds_kernel", "pdm::group_points_rows_kernel")),
                    "launches_per_step": gp["calls_per_step"], "avg_launch_us": round(gp_launch_s * 1e6, 2),
                    "alg_bytes_per_launch": int(gp_launch_bytes)}
    pg = [o for o in ops if o["op"] == "pdm_gather_bev"]
    roofline_pdm = None
    if pg:
        o = pg[0]
        gbs = o["alg_MB_per_step"] / o["ms_per_step"]
        roofline_pdm = {"bound": "hbm", "kernel": "pdm::pdm_bin_kernel + pdm::pdm_gather_reg_kernel (pdm_gather_bev, the inference "
                        "form of the PDM scatter; atomics form under pdm_neck_forms)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy": round(gbs / copy_gbs, 4),
                        "frac_of_guide_copy": round(gbs / HBM_GUIDE_COPY_GBS, 4),
                        "traffic": pmc_traffic(("pdm::pdm_gather_reg_kernel",), together=("pdm::pdm_bin_kernel",)),
                        "traffic_unit": "HBM bytes per call (PMC): bin kernel + gather kernel",
                        "launches_per_step": o["calls_per_step"], "avg_launch_us": round(o["ms_per_step"] * 1e3 / o["calls_per_step"], 2),
                        "alg_bytes_per_launch": int(o["alg_MB_per_step"] * 1e6 / o["calls_per_step"])}
    refops["frac_of_measured_copy"] = round(refops["GBps"] / copy_gbs, 4)
    refops["frac_of_guide_copy"] = round(refops["GBps"] / HBM_GUIDE_COPY_GBS, 4)
    # BASELINE's metric is "frames/sec ...; ball_query HBM GB/s": the target block's figure travels INSIDE `roofline` too (the
    # driver's parsed record keeps `roofline`, `config` and `cpu_baseline` whole and only the names of the other keys)
    if roofline is not None:
        lc = refops.get("levels_concurrent") or {}
        roofline["hbm_target_block"] = {
            "what": "north_star target: the API-exact ball_query + group_points calls of PointNet2MSG's four SA levels (8 + 16 launches "
                    "+ the grid builds a new batch needs) at bs=%d x %d pts, index-exact; bytes = SURVEY D4" % (B, N),
            "alg_MB": refops["alg_MB_per_step"], "ms": refops["ms_per_step"], "GBps": refops["GBps"],
            "frac_of_hbm_peak": refops["frac_of_hbm_peak"], "frac_of_guide_copy": refops["frac_of_guide_copy"],
            "ms_concurrent": lc.get("ms_per_step"), "frac_concurrent": lc.get("frac_of_hbm_peak"),
            "timed_as": refops["timed_as"], "target": 0.60}

    cpu = cpu1 = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(model, N, args.clouds, frames=args.cpu_frames)
        if not args.no_extras:
            cpu1 = cpu_baseline(model, N, args.clouds, frames=args.cpu_frames, threads=1)

    if not args.no_extras and not args.serial:
        del bench
        torch.cuda.empty_cache()
        if world == 1 and (B, N) == (32, 16384):
            # (first of the extras: measured 28.5 ms per step behind the three other benches against 25.6 on its own)
            # configs[3] on this GPU: the bf16-autocast training step of the whole detector (real losses, AdamW), 5 steps
            # after 3 warm-up steps, on a fresh copy of the model (the timed inference model stays in eval mode)
            import copy
            import gc
            gc.collect()              # the Bench objects above hold reference cycles (bound methods): their graphs and stream
            torch.cuda.empty_cache()  # pools are only released by a collection, and a live one costs the train step ~3 ms
            tm = copy.deepcopy(model)
            try:
                tl = train_bench(args, tm, points, B, N, rank, world, local_rank, device, steps=5, warmup=3)
                extras["train_step_bf16"] = {k: tl[k] for k in ("metric", "value", "unit", "ms_per_step", "host_issue_ms_per_step", "steps", "warmup", "dtype", "first_loss", "final_loss")}
                extras["train_step_bf16"]["workload"] = tl["config"]["workload"]
            except Exception as e:   # the inference line must not be lost to a training-side failure: say so instead
                extras["train_step_bf16"] = {"error": f"{type(e).__name__}: {e}"}
            del tm
            torch.cuda.empty_cache()

        with torch.no_grad():
            if args.clouds != "lidar":
                b2 = Bench(model, B, N, "lidar", args.pipeline_depth, device, seed0=4321, graph=not args.no_graph,
                           autotune=not args.no_autotune)
                t = b2.timed(max(10, args.steps), 3)
                n = max(10, args.steps)
                extras["lidar_like_clouds"] = {"ms_per_step": round(t / n * 1e3, 4), "frames_per_s": round(B * n / t, 1),
                                               "verified": verify_or_message(b2),
                                               "workload": f"same step, bs={B} x {N} pts, lidar-like clouds (ring structure + ground "
                                                           "plane: 1.2-5.6 distinct neighbours per ball instead of 1)"}
                del b2
                torch.cuda.empty_cache()
            if (B, N) == (32, 16384):
                b5 = Bench(model, 16, 65536, "lidar", 4, device, seed0=777, graph=not args.no_graph, autotune=not args.no_autotune)
                n = 10
                t = b5.timed(n, 2)
                b5.pipe.check_sampling()
                bd5 = neck(backbone({'batch_size': 16, 'points': b5.batches[0], 'points_per_sample_checked': True}))
                extras["config5_dense_stress"] = {
                    "workload": "configs[4]: 65536 pts/cloud, bs=16, lidar-like, same full forward; level-1 FPS as 3 resumable "
                                "segments of 4 cooperating workgroups per cloud",
                    "ms_per_step": round(t / n * 1e3, 4), "frames_per_s": round(16 * n / t, 1), "launch": b5.mode,
                    "verified": verify_or_message(b5),
                    "pdm_neck_forms": pdm_atomics_section(model, bd5)}
                del b5, bd5
                torch.cuda.empty_cache()
            if (B, N) == (32, 16384):
                # configs[1]'s shape (bs = 8 x 16384 points) through the same full forward: the step is bound by the FPS
                # chain's latency there, not by throughput
                b8 = Bench(model, 8, N, args.clouds, args.pipeline_depth, device, seed0=2468, graph=not args.no_graph,
                           autotune=not args.no_autotune)
                n = max(10, args.steps)
                t = b8.timed(n, 3)
                extras["bs8"] = {"workload": f"configs[1] shape: bs=8 x {N} pts, {args.clouds} clouds, same full forward, depth "
                                             f"{b8.depth}", "ms_per_step": round(t / n * 1e3, 4), "frames_per_s": round(8 * n / t, 1),
                                 "launch": b8.mode, "verified": verify_or_message(b8)}
                del b8
                torch.cuda.empty_cache()
        if (B, N) == (32, 16384) and world == 1:
            # OPT-IN, never the headline: the point head's two stacks with fp32 EMULATED on the bf16 matrix pipe (three bf16
            # pieces per operand, six partial products, fp32 accumulation: csrc/rows_chain_x3.hip) — the fp32 MFMA instruction
            # that bounds the step runs at 1/16 of the bf16 rate on this chip.  Kernel time and error table, then the same
            # pipelined step with head.use_x3 = True (verified against the serial detector running the same kernels).
            try:
                with torch.no_grad():
                    extras["split_bf16_chain"] = split_bf16_section(model, B, N, args, device)
            except Exception as e:
                extras["split_bf16_chain"] = {"error": f"{type(e).__name__}: {e}"}
    # rebuilt if deleted above: only its attributes are needed for the line
    line = {
        "metric": f"frames/sec ({N}-pt clouds, bs={B})", "value": round(frames_per_s, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[2]: full PDM-SSD forward (PointNet2MSG backbone + PDM neck + hybrid head: BEV heat-map "
                               f"head + point box head with decoded boxes), bs={B}/GPU x {N} pts, {args.clouds} KITTI-range clouds, "
                               f"fp32 inference, inputs resident in HBM, {Bench.NBATCH} distinct batches in rotation",
                   "launch": LAUNCH_MODE[0], "parallelism": f"dp{world}", "global_batch": world * B,
                   "overlap": "none" if args.serial else
                              ("FPS chain of batch i+1 on a side stream under the feature half of batch i "
                               "(pdm_ssd_amd/pipeline.py)" if args.pipeline_depth == 1 else
                               f"level-1 FPS cut into {args.pipeline_depth - 1} resumable segments: one launch per step runs "
                               f"segment s of batch i+{args.pipeline_depth}-s side by side, the rest of batch i+1's coordinate "
                               "chain on a third stream, under the feature half of batch i (heat-map head behind the neck on "
                               "the neck's stream, point head behind the FP layers); every step does one full batch of every "
                               "kind of work (pdm_ssd_amd/pipeline.py)" if args.pipeline_depth >= 3 else
                               "sampling two batches deep: level-1 FPS of batch i+2 and levels 2-4 of batch i+1 on side "
                               "streams under the feature half of batch i; every step does one full batch of every "
                               "kind of work (pdm_ssd_amd/pipeline.py)"),
                   "ms_per_step_eager_serial": round(serial_ms, 4),
                   "verified": verified,
                   "sa_first_layer_hoisted": LAUNCH_MODE[1]},
        "roofline": roofline,
        "roofline_fps": roofline_fps,
        "roofline_mfma": roofline_mfma,
        "roofline_hbm": roofline_hbm,
        "roofline_pdm": roofline_pdm,
        "hbm_copy_measured_GBps": round(copy_gbs, 1),
        "hbm_copy_guide_GBps": HBM_GUIDE_COPY_GBS,
        "traffic_source": pmc_src,
        "mlp_flops": flop_summary,
        "mfma_entry_points": mfma_blocks,
        "ball_query_plus_group": refops,
        "pdm_neck_forms": pdm,
        "ops": ops,
        "ops_in_situ": insitu,
        "extras": extras,
        "cpu_baseline": cpu,
        "cpu_baseline_single_thread": cpu1,
    }
    print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


LAUNCH_MODE = [None, None]


def split_bf16_section(model, B, N, args, device):
    from pdm_ssd_amd import fused
    from pdm_ssd_amd.dense_heads.point_head_box import _fc_layers
    head = model.point_head
    rows = B * N
    x = torch.randn(rows, 128, device=device)
    stacks = [(head.num_class, head.cls_layers), (head.box_coder.code_size, head.box_layers)]
    p32 = [fused.PackedMLP(_fc_layers(seq), device) for _, seq in stacks]
    px3 = [fused.PackedMLPx3(_fc_layers(seq), device) for _, seq in stacks]
    o32 = [torch.empty(rows, (c + 3) // 4 * 4, device=device) for c, _ in stacks]
    ox3 = [torch.empty(rows, (c + 3) // 4 * 4, device=device) for c, _ in stacks]

    def f32():
        fused.rows_forward_pair(p32[0], p32[1], x, o32[0], o32[1], relu_last=False)

    def x3():
        for p, o in zip(px3, ox3):
            fused.rows_forward_x3(p, x, o, relu_last=False)

    def timed(f, n=20):
        for _ in range(40):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    ms32, msx3 = timed(f32), timed(x3)
    flop = rows * sum(p.flops_per_position for p in p32)
    sub = 32768
    errs = {}
    for (c, seq), a, b in zip(stacks, o32, ox3):
        import copy
        want = copy.deepcopy(seq).cpu().double().eval()(x[:sub].cpu().double())
        scale = float(want.abs().max())
        a64, b64 = a[:sub, :c].cpu().double(), b[:sub, :c].cpu().double()
        errs[f"{c}_outputs"] = {"max_rel_x3_vs_fp32_mfma": float((a64 - b64).abs().max()) / scale,
                                "max_rel_fp32_mfma_vs_float64": float((a64 - want).abs().max()) / scale,
                                "max_rel_x3_vs_float64": float((b64 - want).abs().max()) / scale,
                                "rms_rel_fp32_mfma_vs_float64": float((a64 - want).pow(2).mean().sqrt()) / scale,
                                "rms_rel_x3_vs_float64": float((b64 - want).pow(2).mean().sqrt()) / scale}
    out = {"dtype": "f32 emulated (3 x bf16 split, 6 products on v_mfma_f32_16x16x32_bf16, f32 accumulation)",
           "status": "opt-in (PointHeadBox.use_x3); NOT used by the headline step",
           "workload": f"the point head's two stacks 128 -> 256 -> 256 -> {{{stacks[0][0]}, {stacks[1][0]}}} over {rows} rows",
           "ms_fp32_mfma_one_launch": round(ms32, 4), "ms_split_bf16_two_launches": round(msx3, 4), "speedup": round(ms32 / msx3, 3),
           "TFLOPs_fp32_mfma": round(flop / ms32 / 1e9, 1), "TFLOPs_fp32_equivalent_split_bf16": round(flop / msx3 / 1e9, 1),
           "errors_relative_to_output_scale": errs, "error_sample_rows": sub}
    del x, o32, ox3
    torch.cuda.empty_cache()
    head.use_x3 = True
    try:
        bx = Bench(model, B, N, args.clouds, args.pipeline_depth, device, seed0=1234, graph=not args.no_graph, autotune=not args.no_autotune)
        n = max(10, args.steps)
        t = bx.timed(n, 3)
        out["full_step_with_split_bf16_point_head"] = {"ms_per_step": round(t / n * 1e3, 4), "frames_per_s": round(B * n / t, 1),
                                                        "verified": verify_or_message(bx)}
        del bx
    finally:
        head.use_x3 = False
        torch.cuda.empty_cache()
    return out


def verify_or_message(b, steps=2):
    """`verified` block of an EXTRA configuration: a failure there is reported in its block (the headline's own check
    exits non-zero instead)."""
    try:
        return b.verify(steps)
    except AssertionError as e:
        return {"vs_serial": "FAILED", "error": str(e)}


def mlp_flops_linear(seq):
    return 2 * sum(m.in_features * m.out_features for m in seq if isinstance(m, torch.nn.Linear))


if __name__ == "__main__":
    main()
