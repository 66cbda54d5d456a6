"""A/B of fused-kernel knobs on the bench model (sums of SA and FP kernel times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pdm_ssd_amd import _native
dev = torch.device("cuda:0")
backbone, neck = bench.build_models(dev)
_, points = bench.make_batch(32, 16384, "uniform", 1234, dev)
lib = _native.lib()
def run():
    return backbone({'batch_size': 32, 'points': points, 'points_per_sample_checked': True})
def measure():
    with torch.no_grad():
        run(); torch.cuda.synchronize()
        with bench.OpTimer() as t:
            for _ in range(3):
                run()
            ops = {o["op"]: o["ms_per_step"] for o in t.summary(3)}
    return ops.get("pdm_sa_mlp_fused"), ops.get("pdm_fp_mlp_fused")
for tiles in (1, 2, 1, 2):
    lib.pdm_tune_fused_tiles(tiles)
    sa, fp = measure()
    print(f"tiles={tiles}: SA {sa:.3f} ms  FP {fp:.3f} ms")
lib.pdm_tune_fused_tiles(0)
