"""Diagnostic: how many of the 16 waves run their distance pass per FPS iteration (sample 0), by 256-iteration bucket."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pdm_ssd_amd import synthetic
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfps_diag3.so"))
f = lib.pdm_furthest_point_sampling
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
for kind in ("uniform", "lidar"):
    gen = synthetic.uniform_clouds if kind == "uniform" else synthetic.lidar_like_clouds
    B, N, m = 2, 16384, 4096
    xyz = torch.from_numpy(np.ascontiguousarray(gen(B, N)[:, :, :3])).to(dev)
    temp = torch.full((B, N), 1e10, device=dev)
    # last 16 floats of temp (sample 1's tail) are abused as counters: preset them to 0
    temp.view(-1)[-16:] = 0
    idx = torch.empty((B, m), dtype=torch.int32, device=dev)
    f(0, B, N, m, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr())
    torch.cuda.synchronize()
    c = temp.view(-1)[-16:].flip(0).cpu().numpy()
    print(kind, "computing waves per iteration, per 256-iteration bucket:", np.round(c / 256, 2))
