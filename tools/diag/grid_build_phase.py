"""Where the search-grid build (one 1024-thread workgroup per cloud) spends its time: s_memtime phase stamps of the
BQG_DIAG build (make -C pdm_ssd_amd/csrc diag-bqg -> tools/diag/libbqg_diag.so) and the launch's HIP-event time,
for the two grids a PointNet2MSG step builds for ball queries (SA1: 16384 points, SA2: 4096) at bs = 32."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pdm_ssd_amd import _native, synthetic
diag = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbqg_diag.so")
if os.path.exists(diag) and "--product" not in sys.argv:
    _native.LIB_PATH = diag
print("library:", _native.LIB_PATH, flush=True)
dev = torch.device("cuda:0"); l = _native.lib()
l.pdm_tune_bq_quad(3)      # the build measures points per occupied cell only for the by-density query form
B = 32
for kind in ("uniform", "lidar"):
    gen = synthetic.uniform_clouds if kind == "uniform" else synthetic.lidar_like_clouds
    for n, r in ((16384, 0.1), (4096, 0.5), (16384, 0.0), (1024, 1.0)):
        xyz = torch.from_numpy(np.ascontiguousarray(gen(B, n, 7)[:, :, :3])).to(dev)
        nbytes = l.pdm_ball_query_grid_workspace_bytes(B, n)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3): _native.call("pdm_grid_build", st, B, n, r, xyz.data_ptr(), ws.data_ptr(), nbytes)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): _native.call("pdm_grid_build", st, B, n, r, xyz.data_ptr(), ws.data_ptr(), nbytes)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        base = (ws.data_ptr() + 15) // 16 * 16 - ws.data_ptr()
        hdr = ws[base + B * n * 16: base + B * n * 16 + B * 64].view(torch.float32).view(B, 16).cpu().numpy()
        ncells = hdr[:, 4:8].copy().view(np.int32)
        ph = hdr[:, 9:13].mean(0) / 2.4e3      # us at 2.4 GHz
        print(f"{kind:8s} n={n:6d} r={r:3.1f}: launch {us:6.1f} us | stamps (us @2.4GHz): bbox+h {ph[0]:5.1f}  hist {ph[1]-ph[0]:5.1f}  "
              f"scan {ph[2]-ph[1]:5.1f}  scatter {ph[3]-ph[2]:5.1f}  | grid {ncells[0][:3]} = {ncells[0][3]} cells | points per occupied cell "
              f"{hdr[:, 8].min():.2f} .. {hdr[:, 8].max():.2f}", flush=True)
