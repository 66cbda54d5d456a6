"""clock (s_memtime) phase table of one FPS iteration, from the FPS_DIAG=4 build (make -C pdm_ssd_amd/csrc diag):
waves that ran a distance pass in an iteration vs waves that skipped, workgroup 0, mean ticks per iteration."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pdm_ssd_amd import synthetic
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfps_diag4.so"))
f = lib.pdm_furthest_point_sampling
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
NAMES = ["bound+ballot", "passes", "record write", "barrier wait", "record read", "decode", "iteration"]
B = 32
cases = [(16384, 4096, (4, 5, 7)), (4096, 1024, (17, 19))]
print("| cloud | N | variant | waves | " + " | ".join(NAMES) + " | share of iterations |")
print("|---|---|---|---|" + "---|" * (len(NAMES) + 1))
for kind in ("uniform", "lidar"):
    gen = synthetic.uniform_clouds if kind == "uniform" else synthetic.lidar_like_clouds
    for N, m, variants in cases:
        xyz = torch.from_numpy(np.ascontiguousarray(gen(B, N, 5)[:, :, :3])).to(dev)
        for v in variants:
            lib.pdm_tune_fps_variant(v)
            buf = (ctypes.c_ulonglong * 16)()
            lib.pdm_fps_phase_read(buf, 1)
            temp = torch.full((B, N), 1e10, device=dev)
            idx = torch.empty((B, m), dtype=torch.int32, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f(0, B, N, m, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            lib.pdm_fps_phase_read(buf, 1)
            a = np.array(list(buf), dtype=np.float64).reshape(2, 8)
            tot = a[:, 7].sum()
            for row, name in ((1, "passing"), (0, "skipping")):
                if a[row, 7] > 0:
                    mean = a[row, :7] / a[row, 7]
                    print(f"| {kind} | {N} | {v} | {name} | " + " | ".join(f"{x:.0f}" for x in mean) + f" | {a[row, 7] / tot:.2f} |")
            print(f"| {kind} | {N} | {v} | call | {e0.elapsed_time(e1) * 1e3 / (m - 1):.3f} us/iteration with stamps | | | | | | | |", flush=True)
lib.pdm_tune_fps_variant(0)
