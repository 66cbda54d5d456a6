"""Timing-only diagnostic builds of the fused MLP kernels (results wrong by construction):
diag1 = weights always the same L1-resident fragments; diag2 = no LDS/global B refetch in the K loop;
diag3 = no MFMA (VALU stand-in); diag4 = diag1 + diag2 (matrix work with no operand traffic in the K loop).  Runs the bench model's backbone with the product library swapped."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pdm_ssd_amd import _native
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device("cuda:0")
backbone, neck = bench.build_models(dev)
_, points = bench.make_batch(32, 16384, "uniform", 1234, dev)
full = _native.lib()
orig_call = _native.call

def run():
    return backbone({'batch_size': 32, 'points': points, 'points_per_sample_checked': True})

def measure():
    with torch.no_grad():
        run(); torch.cuda.synchronize()
        with bench.OpTimer() as t:
            for _ in range(3):
                run()
            return {o["op"]: o["ms_per_step"] for o in t.summary(3)}

print("full ", {k: v for k, v in measure().items() if "fused" in k})
for d in (1, 2, 3, 4):
    lib = ctypes.CDLL(os.path.join(here, f"libfused_diag{d}.so"))
    def call(name, stream, *args, _lib=lib):
        if "mlp_fused" in name:
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = getattr(full, name).argtypes
            rc = fn(stream, *args)
            assert rc == 0, rc
        else:
            orig_call(name, stream, *args)
    _native.call = call
    print(f"diag{d}", {k: v for k, v in measure().items() if "fused" in k})
    _native.call = orig_call
