"""Timing-only diagnostic builds of the FPS kernel (results are wrong by construction):
diag1 = no cross-wave exchange/barrier, diag2 = no distance pass.  Shares shapes with the bench."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native
here = os.path.dirname(os.path.abspath(__file__))
libs = {"full": _native.lib()}
for d in (1, 2):
    libs[f"diag{d}"] = ctypes.CDLL(os.path.join(here, f"libfps_diag{d}.so"))
dev = torch.device("cuda:0")
for (N, m) in [(16384, 4096), (4096, 1024), (256, 64)]:
    xyz = torch.rand(32, N, 3, device=dev) * 50
    for name, lib in libs.items():
        f = lib.pdm_furthest_point_sampling
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
        temp = torch.full((32, N), 1e10, device=dev)
        idx = torch.empty((32, m), dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        for variant in ((0, 1) if N == 16384 and name == "full" else (0,)):
            if name == "full":
                lib.pdm_tune_fps_variant(variant)
            f(s, 32, N, m, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f(s, 32, N, m, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr())
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(f"N={N:6d} m={m:5d} {name:6s} v{variant}: {ms:7.3f} ms  {ms*1e3/(m-1):.3f} us/iter")
libs["full"].pdm_tune_fps_variant(0)
