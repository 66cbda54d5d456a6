"""Where the branches of the captured step really run (no profiler): device timestamps (pdm_mark_time) at the start and
end of the FPS launch, the rest of the coordinate chain, the SA stack, the FP stack and the neck, read after replays."""
import sys
sys.path.insert(0, '.')
import torch
import bench
from pdm_ssd_amd import _native
from pdm_ssd_amd.pipeline import PipelinedHotPath, _flat
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils

kind = sys.argv[1] if len(sys.argv) > 1 else 'uniform'
dev = torch.device('cuda:0')
B, N, depth = 32, 16384, 3
backbone, neck = bench.build_models(dev)
_, points = bench.make_batch(B, N, kind, 1234, dev)
backbone.autotune_hoisting(points, B)
pipe = PipelinedHotPath(backbone, neck, depth=depth)
names = ["step0", "fps0", "fps1", "tail0", "tail1", "main0", "sa1", "fp1", "neck0", "neck1", "join", "end"]
marks = torch.zeros(len(names), dtype=torch.int64, device=dev)
def mark(name):
    _native.call("pdm_mark_time", torch.cuda.current_stream().cuda_stream, marks.data_ptr() + 8 * names.index(name))

def step():
    S, nlev = pipe.nseg, len(backbone.SA_modules)
    main = torch.cuda.current_stream()
    m = backbone.SA_modules[0].npoint
    bounds = pipe._bounds()
    ahead = [points] * (S + 1)
    mark("step0")
    pipe.side.wait_stream(main); pipe.side3.wait_stream(main)
    with torch.cuda.stream(pipe.side):
        xyz0 = pipe._xyz(ahead[S], B)
        fresh = pipe._fresh(xyz0, m)
        jobs = [(xyz0, fresh[0], fresh[1], bounds[0], bounds[1])]
        for s in range(1, S):
            jobs.append((pipe._xyz(ahead[S - s], B), pipe.seg[s][0], pipe.seg[s][1], bounds[s], bounds[s + 1]))
        mark("fps0"); pointnet2_utils.fps_segments(jobs, m); mark("fps1")
    with torch.cuda.stream(pipe.side3):
        mark("tail0")
        nxt = backbone.coordinate_levels(pipe._xyz(ahead[0], B), 0, nlev, first_idx=pipe.l1idx)
        mark("tail1")
    bd = {'batch_size': B, 'points': points, 'points_per_sample_checked': True}
    bd.update(pipe.cur)
    def start_neck(d):
        mark("sa1")
        pipe.neck_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(pipe.neck_stream):
            mark("neck0"); neck(d); mark("neck1")
    bd['after_sa_hook'] = start_neck
    mark("main0")
    bd = backbone(bd)
    mark("fp1")
    main.wait_stream(pipe.neck_stream); main.wait_stream(pipe.side3); main.wait_stream(pipe.side)
    mark("join")
    _native.copy_many(_flat(pipe.cur), _flat(nxt))
    _native.copy_many([pipe.l1idx], [pipe.seg[S - 1][1]])
    for s in range(S - 1, 1, -1):
        _native.copy_many(list(pipe.seg[s]), list(pipe.seg[s - 1]))
    _native.copy_many(list(pipe.seg[1]), list(fresh))
    mark("end")
    return bd['spatial_features'], bd['point_features']

with torch.no_grad():
    pipe.prime_segmented([points] * depth, B)
    for _ in range(2): step()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): step()
    torch.cuda.current_stream().wait_stream(s)
    for mode in ("hipGraph", "eager"):
        if mode == "hipGraph":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): out = step()
            run = g.replay
        else:
            run = step
        for _ in range(5): run()
        torch.cuda.synchronize()
        t = marks.cpu().numpy().astype('float64')
        t = (t - t[0]) / 100.0   # us
        print(kind, mode, " ".join(f"{n}={v:.0f}" for n, v in zip(names, t)))
