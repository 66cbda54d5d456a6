"""Experiment: the depth>=3 step as three hipGraphs (coordinates / features / hand-over) on their own streams."""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from pdm_ssd_amd.pipeline import PipelinedHotPath

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
B, N = 32, 16384
backbone, neck = bench.build_models(dev)
_, points = bench.make_batch(B, N, 'uniform', 1234, dev)
pipe = PipelinedHotPath(backbone, neck, depth=depth)
ahead = [points] * depth
extra = {'points_per_sample_checked': True}
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def capture(fn, stream):
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        out = fn()
    return g, out

with torch.no_grad():
    pipe.prime_segmented(ahead, B)
    for _ in range(2):
        pipe.part_coordinates(ahead, B); pipe.part_features(points, B, extra); pipe.part_handover()
    torch.cuda.synchronize()
    gB, _ = capture(lambda: pipe.part_coordinates(ahead, B), sB)
    gA, outA = capture(lambda: pipe.part_features(points, B, extra), sA)
    gH, _ = capture(pipe.part_handover, sA)
    evH, evB = torch.cuda.Event(), torch.cuda.Event()

    def run():
        # previous hand-over (on sA) must be done before the coordinate graph reads/writes the static state
        evH.record(sA) if False else None
        sB.wait_event(evH)
        with torch.cuda.stream(sB):
            gB.replay()
            evB.record(sB)
        with torch.cuda.stream(sA):
            gA.replay()
            sA.wait_event(evB)
            gH.replay()
            evH.record(sA)

    evH.record(sA)
    for _ in range(5): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"depth {depth} split graphs: {dt * 1e3:.4f} ms/step, {B / dt:.1f} frames/s")
    # check against the serial path
    bd = neck(backbone({'batch_size': B, 'points': points, 'points_per_sample_checked': True}))
    torch.cuda.synchronize()
    print("equal:", torch.equal(bd['point_features'], outA['point_features']), torch.equal(bd['spatial_features'], outA['spatial_features']))
