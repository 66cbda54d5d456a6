"""BASELINE's target block (the API-exact ball_query + group_points calls of PointNet2MSG's four SA levels, one shared grid
build per level) as ONE hipGraph replayed many times — nothing else in the process, so that
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_api -- python3 tools/diag/api_block.py [uniform|lidar] [replays]
gives every kernel's duration INSIDE the sequence (cold operands, real launch order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native
if "--lib" in sys.argv:      # an alternative build of the library (e.g. tools/diag/libbq_u8.so)
    i = sys.argv.index("--lib")
    _native.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
    print("library:", _native.LIB_PATH, flush=True)
if "--quad" in sys.argv:     # grid ball query form: 2 lane (default), 1 quad, 0 wave
    i = sys.argv.index("--quad")
    _native.lib().pdm_tune_bq_quad(int(sys.argv[i + 1]))
    del sys.argv[i:i + 2]
CHAINS = 0
if "--chains" in sys.argv:   # 4 = one chain per SA level, 8 = one per (level, scale) behind the level's grid build; streams, fork / join only
    i = sys.argv.index("--chains")
    CHAINS = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
GP_TUNE = None
if "--gp-tune" in sys.argv:  # pdm_tune_group_rows word (tools/diag/group_sweep.py)
    i = sys.argv.index("--gp-tune")
    GP_TUNE = int(sys.argv[i + 1], 0)
    del sys.argv[i:i + 2]
GP_FLOOR = None
if "--gp-lds-floor" in sys.argv:
    i = sys.argv.index("--gp-lds-floor")
    GP_FLOOR = int(sys.argv[i + 1], 0)
    del sys.argv[i:i + 2]
if "--cpw" in sys.argv:
    i = sys.argv.index("--cpw")
    _native.lib().pdm_tune_bq_cpw(int(sys.argv[i + 1]))
    del sys.argv[i:i + 2]
if "--heavy" in sys.argv:
    i = sys.argv.index("--heavy")
    _native.lib().pdm_tune_bq_heavy(int(sys.argv[i + 1]))
    del sys.argv[i:i + 2]
import bench
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 50
backbone, neck = bench.build_models(dev)
B = 32
_, points = bench.make_batch(B, 16384, kind, 1234, dev)
xyz = points[:, 1:4].contiguous().view(B, -1, 3)
feats = points[:, 4:].contiguous().view(B, -1, 1).permute(0, 2, 1).contiguous()
chans = [1, 96, 256, 512]
plan, totb = [], 0
with torch.no_grad():
    for k, m in enumerate(backbone.SA_modules):
        new_xyz = m.sample(xyz)
        f = feats if k == 0 else torch.randn(B, chans[k], xyz.shape[1], device=dev)
        xt = xyz.transpose(1, 2).contiguous()
        N, M = xyz.shape[1], new_xyz.shape[1]
        for g in m.groupers:
            plan.append((g.radius, g.nsample, xyz, new_xyz, f, xt))
            totb += B * (12 * N + 12 * M + 4 * M * g.nsample)
            for C in (3, f.shape[1]):
                totb += B * (4 * M * g.nsample + 4 * C * N + 4 * C * M * g.nsample)
        xyz = new_xyz
    if GP_FLOOR is not None:
        _native.lib().pdm_tune_group_lds_floor(GP_FLOOR)
    if GP_TUNE is not None:
        _native.lib().pdm_tune_group_rows(GP_TUNE)
    streams = [torch.cuda.Stream() for _ in range(8)]
    def whole():
        if not CHAINS:
            with pu.shared_search_grids():
                for radius, ns, x, nx, f, xt in plan:
                    idx = pu.ball_query(radius, ns, x, nx)
                    pu.grouping_operation(xt, idx)
                    pu.grouping_operation(f, idx)
            return
        cur = torch.cuda.current_stream()
        def scale(entry, big=True):
            radius, ns, x, nx, f, xt = entry
            idx = pu.ball_query(radius, ns, x, nx)
            pu.grouping_operation(xt, idx)
            if big or f.shape[1] == 1:
                pu.grouping_operation(f, idx)
                return None
            return (f, idx)
        for st in streams:
            st.wait_stream(cur)
        later = []
        with pu.shared_search_grids(cross_stream=True):
            for li in range(4):
                a, b2 = streams[2 * li], streams[2 * li + 1]
                if CHAINS == 4:
                    with torch.cuda.stream(a):
                        scale(plan[2 * li]); scale(plan[2 * li + 1])
                    continue
                if CHAINS in (2, 3):
                    # 2: level 1 (grid build, searches, small copies: latency-bound, 0.1 of the 1.68 GB) on one stream, levels 2-4 (the
                    #    streaming copies) one after the other on a second.  3: levels 1 | 2 | 3 + 4.
                    st = streams[0] if li == 0 else streams[1] if (CHAINS == 2 or li == 1) else streams[2]
                    with torch.cuda.stream(st):
                        scale(plan[2 * li]); scale(plan[2 * li + 1])
                    continue
                # 8 chains: the level's grid build + first search on a, the second scale behind the build only.
                # 9 = the same, and the six feature gathers of levels 2-4 (the streaming part: 1.2 of 1.68 GB) AFTER every chain
                # has been joined, one after the other: searches and small copies no longer wait for slots beside them.
                radius, ns, x, nx, f, xt = plan[2 * li]
                with torch.cuda.stream(a):
                    idx = pu.ball_query(radius, ns, x, nx)
                b2.wait_stream(a)
                with torch.cuda.stream(a):
                    pu.grouping_operation(xt, idx)
                    if CHAINS == 8 or f.shape[1] == 1:
                        pu.grouping_operation(f, idx)
                    else:
                        later.append((f, idx))
                with torch.cuda.stream(b2):
                    r = scale(plan[2 * li + 1], big=CHAINS == 8)
                    if r: later.append(r)
            for st in streams:          # join only after every chain has been forked
                cur.wait_stream(st)
            for f, idx in later:
                pu.grouping_operation(f, idx)
    whole(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        whole()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        whole()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / replays * 1e3
print(f"{kind}: whole sequence as one graph, {replays} replays: {us:.1f} us -> {totb/1e3/us:.0f} GB/s = {totb/1e3/us/8000:.3f} of 8 TB/s ({totb/1e6:.1f} MB)")
