#!/usr/bin/env python3
"""bench.py — frames/s of PDM-SSD's point-cloud hot path on MI355X.

One "step" = one forward pass of the hot path over one batch of synthetic KITTI-range clouds that are
already resident in HBM: PointNet2MSG backbone (4 SA-MSG + 4 FP layers: FPS, ball query, grouping,
shared MLPs, three-NN interpolation) followed by the PDM neck (dilation, SH x Gaussian filling,
scatter-add to the BEV grid, normalise, height compression).  fp32, inference (BN in eval mode), the
configuration BASELINE.json's metric is quoted on: bs = 32 clouds of 16384 points per GPU.

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
from pdm_ssd_amd.pdm_neck import PDMNeck
from pdm_ssd_amd.pipeline import PipelinedHotPath
from pdm_ssd_amd.pointnet2_backbone import POINTRCNN_MSG_CFG, PointNet2MSG

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak (= fp32 vector peak)
VOXEL = [0.05, 0.05, 0.1]
NECK_CFG = {'SOURCE_LAYER': 2, 'FEATURE_DIM': 128, 'DILATION': [7, 7, 1], 'SH_DEGREE': 2, 'BEV_STRIDE': 8,
            'HEIGHT_BINS': 1, 'INPUT_CHANNELS': 256, 'NORMALIZE': True}


def build_models(device, seed=0):
    torch.manual_seed(seed)
    backbone = PointNet2MSG(POINTRCNN_MSG_CFG, input_channels=4)
    neck = PDMNeck(NECK_CFG, grid_size=[1408, 1600, 40], voxel_size=VOXEL,
                   point_cloud_range=list(synthetic.KITTI_RANGE))
    with torch.no_grad():  # non-degenerate SH / scale head so the neck's arithmetic is fully exercised
        neck.coef.weight.normal_(0.0, 0.02)
    return backbone.to(device).eval(), neck.to(device).eval()


def make_batch(B, N, kind, seed0, device):
    gen = synthetic.uniform_clouds if kind == "uniform" else synthetic.lidar_like_clouds
    clouds = gen(B, N, seed0)
    pts = torch.from_numpy(synthetic.to_batch_points(clouds)).to(device)
    return clouds, pts


# ----------------------------------------------------------------------------- per-op accounting

def algorithmic_bytes(name, a):
    """SURVEY.md section 8 D4 formulas; `a` = the integer/float arguments of the C-ABI call."""
    if name in ("pdm_ball_query", "pdm_ball_query_grid"):
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
    if name in ("pdm_three_nn", "pdm_three_nn_grid"):
        b, n, m = a[:3]
        return b * (12 * n + 12 * m + 24 * n)
    if name == "pdm_three_interpolate":
        b, c, m, n = a[:4]
        return b * (24 * n + 4 * c * m + 4 * c * n)
    if name == "pdm_scatter_bev":
        B, P, C, deg = a[:4]
        W, H, D = a[17:20]
        return B * (12 * P + 4 * C * P + 4 * (deg + 1) ** 2 * P + 4 * P) + 4 * B * C * D * H * W
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
            d["bytes"] += algorithmic_bytes(name, ints)
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
    with OpTimer() as t:
        for it in range(iters + 1):
            if it == 1:
                t.records.clear()  # first pass = warm-up
            for radius, ns, x, nx, f, xt in plan:
                idx = pu.ball_query(radius, ns, x, nx)
                pu.grouping_operation(xt, idx)
                pu.grouping_operation(f, idx)
        ops = t.summary(iters)
    ms = sum(o["ms_per_step"] for o in ops)
    mb = sum(o["alg_MB_per_step"] for o in ops)
    gp = [o for o in ops if o["op"] == "pdm_group_points"][0]
    return {"ops": ops, "ms_per_step": round(ms, 4), "alg_MB_per_step": round(mb, 2),
            "GBps": round(mb / ms, 1), "frac_of_hbm_peak": round(mb / ms / HBM_PEAK_GBS, 4),
            "note": "8 ball_query + 16 group_points launches at bs=%d; target >= 0.60" % B}, gp


# ----------------------------------------------------------------------------- CPU baseline

def cpu_baseline(backbone, neck, N, kind, frames=2):
    """Times the CPU statement of the same step (oracle operators + torch-CPU MLPs) on `frames` clouds."""
    import copy

    from oracle import cpu_backbone, cpu_oracle
    cpu_oracle.build()
    bb = copy.deepcopy(backbone).cpu().eval()
    nk = copy.deepcopy(neck).cpu().eval()
    gen = synthetic.uniform_clouds if kind == "uniform" else synthetic.lidar_like_clouds
    clouds = gen(frames, N, 4321)
    threads = cpu_oracle.max_threads()
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    out = cpu_backbone.backbone_forward(bb, clouds)
    cpu_backbone.neck_forward(nk, out['sa_xyz'], out['sa_features'])
    dt = time.perf_counter() - t0
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{frames} clouds x {N} pts, same step (oracle C operators with OpenMP + torch-CPU MLPs), "
                      f"{dt:.1f} s wall; the reference itself has no CPU path for these operators"}


def train_bench(args, backbone, neck, points, B, N, rank, world, local_rank, device):
    """Training step (BASELINE config 4): the autograd graph over the HIP operators (fused QueryAndGroup forward,
    atomic scatter backward kernels), torch modules for the MLPs under bf16 autocast, coordinates and indices in
    fp32; one gradient all-reduce per step (DDP, single bucket) when world > 1.  The loss is a stand-in (mean
    square of both outputs): the reference's hybrid head is not part of the hot path."""
    model = torch.nn.ModuleDict({"backbone": backbone, "neck": neck}).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3)

    class Step(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, pts, sampled=None):
            bd = {'batch_size': B, 'points': pts, 'points_per_sample_checked': True}
            if sampled is not None:
                bd['sampled_xyz'] = sampled
            bd = self.m["neck"](self.m["backbone"](bd))
            sf = bd['spatial_features']
            if not sf.is_contiguous() and sf.permute(0, 2, 3, 1).is_contiguous():
                sf = sf.permute(0, 2, 3, 1)   # the neck's grid is channels-last storage: same mean, contiguous kernels
            return bd['point_features'].float().square().mean() + sf.float().square().mean()

    stepper = Step(model)
    if world > 1:
        stepper = torch.nn.parallel.DistributedDataParallel(stepper, device_ids=[local_rank], bucket_cap_mb=64,
                                                            gradient_as_bucket_view=True)

    # The sampling chain (FPS + gather: 4.4 ms, one workgroup per cloud) needs no gradient: the chain of the NEXT batch
    # runs on a side stream under this batch's forward/backward (same synthetic cloud every step).
    side = torch.cuda.Stream()
    state = {"sampled": None}

    def sample_next():
        with torch.no_grad():
            return backbone.sample_chain(points[:, 1:4].contiguous().view(B, -1, 3))

    if not args.serial:
        state["sampled"] = sample_next()

    def step():
        opt.zero_grad(set_to_none=True)
        if args.serial:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = stepper(points)
        else:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                nxt = sample_next()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = stepper(points, state["sampled"])
        loss.backward()
        opt.step()
        if not args.serial:
            main.wait_stream(side)
            for t in nxt:
                t.record_stream(main)
            state["sampled"] = nxt
        return loss

    for _ in range(max(1, args.warmup)):
        step()
    dist_utils.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    dist_utils.barrier()
    elapsed = dist_utils.max_over_ranks(time.perf_counter() - t0, device)
    if rank == 0:
        print(json.dumps({
            "metric": f"train frames/sec ({N}-pt clouds, bs={B}/GPU, bf16 autocast)", "value": round(world * B * args.steps / elapsed, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 (MLPs) / f32 (coordinates, operators)", "data": "synthetic",
            "config": {"workload": f"configs[3]: train step of PointNet2MSG + PDM neck, bs={B}/GPU x {N} pts, stand-in loss, "
                                   "AdamW, DDP gradient all-reduce over RCCL", "parallelism": f"dp{world}",
                       "overlap": "none" if args.serial else "FPS chain of the next batch on a side stream"},
            "final_loss": float(loss.detach())}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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


# ----------------------------------------------------------------------------- main

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--clouds", choices=["uniform", "lidar"], default="uniform")
    ap.add_argument("--pipeline-depth", type=int, default=3, choices=[1, 2, 3, 4, 5],
                    help="batches whose sampling chain is in flight beside the feature path; >= 3 also cuts the level-1 "
                         "FPS into depth - 1 resumable segments run side by side (pdm_ssd_amd/pipeline.py)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-autotune", action="store_true",
                    help="keep the first SA layer of every level hoisted instead of choosing per level from the "
                         "neighbour density of the batch (PointNet2MSG.autotune_hoisting)")
    ap.add_argument("--serial", action="store_true", help="no cross-batch overlap of the FPS chain")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 4 instead: bf16-autocast forward+backward+AdamW step of backbone+neck, "
                         "DistributedDataParallel gradient all-reduce over RCCL when --gpus > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
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

    B, N = args.batch, args.points
    backbone, neck = build_models(device)
    _, points = make_batch(B, N, args.clouds, 1234 + rank * B, device)
    if args.train:
        return train_bench(args, backbone, neck, points, B, N, rank, world, local_rank, device)

    def step_serial():
        bd = {'batch_size': B, 'points': points, 'points_per_sample_checked': True}
        bd = backbone(bd)
        bd = neck(bd)
        return bd['spatial_features'], bd['point_features']

    fps_wgs = (args.pipeline_depth - 1) * B * ((N + 16383) // 16384)
    if args.pipeline_depth >= 3 and not (1024 < N <= 131072 and
                                         (N <= 16384 or fps_wgs <= _native.lib().pdm_fps_max_coresident_workgroups())):
        print(f"[bench] resumable FPS segments need 1024 < points <= 131072 and co-resident workgroups; "
              f"{N} points x {B} clouds -> --pipeline-depth 2", file=sys.stderr)
        args.pipeline_depth = 2
    pipe = PipelinedHotPath(backbone, neck, depth=args.pipeline_depth)

    def step_pipelined():
        # features of this batch || sampling of the following batch(es) (same synthetic cloud every step)
        bd = pipe.step(points, points, B, extra={'points_per_sample_checked': True}, points_next2=points,
                       points_ahead=[points] * args.pipeline_depth)
        return bd['spatial_features'], bd['point_features']

    step = step_serial if args.serial else step_pipelined

    barrier = dist_utils.barrier

    mode = "eager"
    with torch.no_grad():
        # per-sample point-count check of the backbone (host sync) done once, outside the timed region
        counts = torch.bincount(points[:, 0].long(), minlength=B)
        assert int(counts.min()) == int(counts.max()) == N
        if not args.no_autotune:
            hoisting = backbone.autotune_hoisting(points, B)
        if args.pipeline_depth >= 3:
            pipe.prime_segmented([points] * args.pipeline_depth, B)
        else:
            pipe.prime(points, B)
        for _ in range(max(1, args.warmup)):
            step()
        torch.cuda.synchronize()
        run = step
        graph = None
        if not args.no_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    step()
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = step()
                run = graph.replay
                mode = "hipGraph"
                run()
                torch.cuda.synchronize()
            except Exception as e:  # capture unsupported -> measure eager, say so
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                graph, run, mode = None, step, "eager"
                torch.cuda.synchronize()

        for _ in range(args.warmup):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        barrier()
        elapsed = time.perf_counter() - t0

        elapsed = dist_utils.max_over_ranks(elapsed, device)
        if not args.serial:
            pipe.check_sampling()   # N > 16384: no cooperating FPS workgroup gave up waiting for a peer

        # per-kernel pass (eager, instrumented with HIP events on the launch stream); rank 0 only
        ops = []
        if rank == 0:
            psteps = max(3, min(args.steps, 10))
            from pdm_ssd_amd import fused as _fused
            _fused.FLOP_COUNTER = {}
            with OpTimer() as timer:
                for _ in range(psteps):
                    step_serial()
                ops = timer.summary(psteps)
            executed = {k: v / psteps for k, v in _fused.FLOP_COUNTER.items()}
            _fused.FLOP_COUNTER = None
            t0s = time.perf_counter()
            for _ in range(psteps):
                step_serial()
            torch.cuda.synchronize()
            serial_ms = (time.perf_counter() - t0s) / psteps * 1e3

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    frames_per_s = world * B * args.steps / elapsed

    # HBM traffic per launch from the committed PMC passes of this same command (profiles/r01c_pmc_traffic.json;
    # rocprofv3 cannot run inside this process): corrected bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB
    pmc = {}
    try:
        import glob as _glob
        pmc = json.load(open(sorted(_glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]))["kernels"]
    except Exception:
        pass

    def pmc_traffic(*prefixes):
        rows = [v for k, v in pmc.items() if k.startswith(prefixes) and v["launches"] > 0]
        if not rows:
            return None
        return int(sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in rows) / sum(v["launches"] for v in rows))

    # FLOPs: `executed` = what each entry point actually contracts (2 * real cin * cout per position and layer).
    # The SA / FP kernels run with the wide block of their first layer hoisted onto the source / known points
    # (pdm_rows_mlp_fused), so the step executes fewer FLOPs than the reference's form of the same network.
    sa_gf, fp_gf = model_flops(backbone, B, N)
    for o in ops:
        if o["op"] in executed:
            o["executed_GFLOP_per_step"] = round(executed[o["op"]] / 1e9, 2)
            o["TFLOPs"] = round(executed[o["op"]] / 1e9 / o["ms_per_step"], 1)
    mlp_ops = [o for o in ops if o["op"] in executed]
    flop_summary = {"reference_form_GFLOP_per_step": round(sa_gf + fp_gf, 2),
                    "executed_GFLOP_per_step": round(sum(executed.values()) / 1e9, 2),
                    "mlp_kernels_ms_per_step": round(sum(o["ms_per_step"] for o in mlp_ops), 4),
                    "note": "SA kernels run over compacted neighbour lists (ball_query's padding copies of the first hit are "
                            "not computed: max-pool over a multiset = over the set; bit-identical; uniform clouds hold one point "
                            "per ball, lidar-like ones 1.2-5.6); first-layer hoisting where it pays: W1 [f_nb ; dx] = (W1f f)[nb] "
                            "+ W1x dx (SA, chosen per level by autotune_hoisting), W1 interp(f) = interp(W1 f) (FP)"}
    # dominant roofline-bounded kernel of the step: the fused SA kernel (fp32 MFMA), all its launches (both entry
    # points).  FPS takes longer but is a latency-bound dependency chain (one workgroup per cloud) with no
    # bandwidth or matrix roofline; it is listed under "ops" with its iteration rate.
    # (with the neighbour lists compacted the SA kernels only run the distinct neighbours — on sparse clouds a small
    # part of the step — so the dominant MFMA entry point is picked by measured time, not by name)
    KERNELS_OF = {"pdm_sa_mlp_fused": ("pdm::sa_mlp_fused_kernel", "pdm::sa_reg_mlp_kernel"),
                  "pdm_sa_mlp_fused_pre": ("pdm::sa_mlp_fused_kernel",),
                  "pdm_sa_mlp_packed": ("pdm::sa_packed_fused_kernel", "pdm::sa_reg_packed_kernel"),
                  "pdm_fp_mlp_fused": ("pdm::fp_mlp_fused_kernel",),
                  "pdm_fp_mlp_fused_pre": ("pdm::fp_mlp_fused_kernel", "pdm::rows_gemm_kernel<true>"),
                  "pdm_rows_mlp_fused": ("pdm::fp_mlp_fused_kernel", "pdm::rows_gemm_kernel<false>")}
    sa_ops = sorted(mlp_ops, key=lambda o: -o["ms_per_step"])[:1]
    roofline = None
    if sa_ops:
        calls = sum(o["calls_per_step"] for o in sa_ops)
        per_launch_flop = sum(executed[o["op"]] for o in sa_ops) / calls
        per_launch_s = sum(o["ms_per_step"] for o in sa_ops) / 1e3 / calls
        ach = per_launch_flop / per_launch_s / 1e12
        kernels = KERNELS_OF.get(sa_ops[0]["op"], ())
        roofline = {"bound": "mfma", "kernel": " + ".join(kernels) + f" (all launches of {sa_ops[0]['op']}, the MFMA entry "
                                               "point with the most time in the step)",
                    "achieved": round(ach, 2),
                    "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                    "traffic": pmc_traffic(*kernels) if kernels else None, "traffic_unit": "HBM bytes per launch (PMC)",
                    "launches_per_step": calls,
                    "avg_launch_us": round(per_launch_s * 1e6, 2), "alg_flop_per_launch": int(per_launch_flop),
                    "flops_counted": "executed (hoisted first layer), unpadded"}
    with torch.no_grad():
        refops, gp = reference_op_section(backbone, points, B)
    gp_launch_bytes = gp["alg_MB_per_step"] * 1e6 / gp["calls_per_step"]
    gp_launch_s = gp["ms_per_step"] / 1e3 / gp["calls_per_step"]
    roofline_hbm = {"bound": "hbm", "kernel": "pdm::group_points_v4_kernel (pdm_group_points, API-exact operator)",
                    "achieved": round(gp_launch_bytes / gp_launch_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gp_launch_bytes / gp_launch_s / 1e9 / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic("pdm::group_points_v4_kernel"),
                    "launches_per_step": gp["calls_per_step"], "avg_launch_us": round(gp_launch_s * 1e6, 2),
                    "alg_bytes_per_launch": int(gp_launch_bytes)}
    fps_ops = [o for o in ops if o["op"].startswith("pdm_furthest_point_sampling")]
    if fps_ops:
        iters = sum(m.npoint - 1 for m in backbone.SA_modules)
        fps_ms = sum(o["ms_per_step"] for o in fps_ops)
        fps_ops[0]["fps_chain"] = {"serial_iterations_per_step": iters, "ms_per_step": round(fps_ms, 4),
                                   "us_per_iteration": round(fps_ms * 1e3 / iters, 3)}

    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(backbone, neck, N, args.clouds, frames=args.cpu_frames)

    line = {
        "metric": "frames/sec (16384-pt clouds, bs=32)", "value": round(frames_per_s, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[2]: PointNet2MSG backbone + PDM neck forward, bs={B}/GPU x {N} pts, "
                               f"{args.clouds} KITTI-range clouds, fp32 inference, inputs resident in HBM",
                   "launch": mode, "parallelism": f"dp{world}",
                   "overlap": "none" if args.serial else
                              ("FPS chain of batch i+1 on a side stream under the feature half of batch i "
                               "(pdm_ssd_amd/pipeline.py)" if args.pipeline_depth == 1 else
                               f"level-1 FPS cut into {args.pipeline_depth - 1} resumable segments: one launch per step runs "
                               f"segment s of batch i+{args.pipeline_depth}-s side by side, the rest of batch i+1's coordinate "
                               "chain on a third stream, under the feature half of batch i; every step does one full "
                               "batch of every kind of work (pdm_ssd_amd/pipeline.py)" if args.pipeline_depth >= 3 else
                               "sampling two batches deep: level-1 FPS of batch i+2 and levels 2-4 of batch i+1 on side "
                               "streams under the feature half of batch i; every step does one full batch of every "
                               "kind of work (pdm_ssd_amd/pipeline.py)"),
                   "ms_per_step_eager_serial": round(serial_ms, 4),
                   "sa_first_layer_hoisted": None if args.no_autotune else [d["use_pre"] for d in hoisting]},
        "roofline": roofline,
        "roofline_hbm": roofline_hbm,
        "mlp_flops": flop_summary,
        "ball_query_plus_group": refops,
        "ops": ops,
        "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
