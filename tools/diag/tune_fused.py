"""Sweep the fused-kernel tuning knobs on the bench model (per-call sums of SA and FP kernels)."""
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
    bd = {'batch_size': 32, 'points': points, 'points_per_sample_checked': True}
    return backbone(bd)

def measure():
    with torch.no_grad():
        run(); torch.cuda.synchronize()
        with bench.OpTimer() as t:
            for _ in range(3):
                run()
            ops = {o["op"]: o["ms_per_step"] for o in t.summary(3)}
    return ops.get("pdm_sa_mlp_fused"), ops.get("pdm_fp_mlp_fused")

for groups in (0, 1, 2, 4, 8):
    lib.pdm_tune_fused_groups(groups)
    sa, fp = measure()
    print(f"groups={groups} (0=auto): SA {sa:.3f} ms  FP {fp:.3f} ms")
for waves in (1, 2, 4):
    for groups in (0, 1):
        lib.pdm_tune_fused_waves(waves); lib.pdm_tune_fused_groups(groups)
        sa, fp = measure()
        print(f"forced waves={waves} groups={groups}: SA {sa:.3f} ms  FP {fp:.3f} ms")
lib.pdm_tune_fused_waves(0); lib.pdm_tune_fused_groups(0)
