"""What the LAUNCH STRUCTURE of BASELINE's target block costs by itself: the same 26-launch dependent sequence as one replayed
hipGraph, (a) with a trivial kernel per launch, (b) with a streaming copy per launch that moves exactly the launch's algorithmic
bytes (half read, half written, through pdm_copy_many) — the block as a chain of ideal bandwidth kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native
dev = torch.device("cuda:0"); l = _native.lib()
B = 32
levels = [(16384, 4096, 1), (4096, 1024, 96), (1024, 256, 256), (256, 64, 512)]
launches = []          # algorithmic bytes per launch, SURVEY D4
for N, M, C in levels:
    if N >= 2048: launches.append(("grid build", B * (12 * N + 16 * N)))
    for ns in (16, 32):
        launches.append(("ball_query", B * (12 * N + 12 * M + 4 * M * ns)))
        for c in (3, C):
            launches.append(("group_points", B * (4 * M * ns + 4 * c * N + 4 * c * M * ns)))
total = sum(b for n, b in launches if n != "grid build")
big = torch.empty(160 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(big)
tiny_src, tiny_dst = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
def graph_us(fn, replays=40):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / replays * 1e3
def trivial():
    for _ in launches: _native.copy_many([tiny_dst], [tiny_src])
def streaming():
    off = 0
    for name, nbytes in launches:
        n = max(nbytes // 8, 16)        # floats: half of the bytes read, half written
        _native.copy_many([dst[off:off + n]], [big[off:off + n]])
        off += (n + 63) // 64 * 64
for rep in range(2):
    a, b = graph_us(trivial), graph_us(streaming)
    print(f"{len(launches)} dependent launches, trivial kernel each: {a:.1f} us ({a / len(launches):.2f} us per launch) | streaming copies of the "
          f"launches' algorithmic bytes ({total / 1e6:.0f} MB + grid builds): {b:.1f} us = {total / 1e3 / b / 8000:.3f} of 8 TB/s", flush=True)
