"""Dump the captured step graph (bs=32, depth 3) as DOT via hipGraphDebugDotPrint and list the root nodes and the
predecessors of the first feature-path kernels."""
import ctypes, re, sys
sys.path.insert(0, '.')
import torch
import bench
from pdm_ssd_amd.pipeline import PipelinedHotPath

depth = 3
dev = torch.device('cuda:0')
B, N = 32, 16384
backbone, neck = bench.build_models(dev)
_, points = bench.make_batch(B, N, 'uniform', 1234, dev)
pipe = PipelinedHotPath(backbone, neck, depth=depth)
def step():
    bd = pipe.step(points, points, B, extra={'points_per_sample_checked': True}, points_next2=points, points_ahead=[points] * depth)
    return bd['spatial_features'], bd['point_features']
with torch.no_grad():
    pipe.prime_segmented([points] * depth, B)
    for _ in range(2): step()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g): out = step()
    raw = g.raw_cuda_graph()
    print("raw graph handle", raw)
    hip = ctypes.CDLL(None)
    try:
        fn = hip.hipGraphDebugDotPrint
    except AttributeError:
        import glob, os
        path = glob.glob(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so*'))[0]
        fn = ctypes.CDLL(path).hipGraphDebugDotPrint
    fn.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
    rc = fn(ctypes.c_void_p(raw), b"gpurun_out/graph.dot", 0)
    print("hipGraphDebugDotPrint rc", rc)
