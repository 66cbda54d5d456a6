"""CPU cost of one hipGraph replay of the pipelined step (launch call only, GPU idle before) vs the GPU time of a step."""
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
def step():
    bd = pipe.step(points, points, B, extra={'points_per_sample_checked': True}, points_next2=points, points_ahead=[points] * depth)
    return bd['spatial_features'], bd['point_features']
with torch.no_grad():
    if depth >= 3: pipe.prime_segmented([points] * depth, B)
    else: pipe.prime(points, B)
    for _ in range(3): step()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    g.enable_debug_mode()
    with torch.cuda.graph(g): out = step()
    try:
        g.debug_dump("gpurun_out/graph_depth%d.dot" % depth)
    except Exception as e:
        print("debug_dump failed:", e)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    cpu = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        cpu.append((t1 - t0, t2 - t0))
    print("depth", depth, "replay call ms", [round(c[0] * 1e3, 3) for c in cpu])
    print("replay+sync ms", [round(c[1] * 1e3, 3) for c in cpu])
    t0 = time.perf_counter()
    for _ in range(30): g.replay()
    torch.cuda.synchronize()
    print("back-to-back ms/step", (time.perf_counter() - t0) / 30 * 1e3)
