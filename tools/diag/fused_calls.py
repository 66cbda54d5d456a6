"""Per-launch times of the fused MLP entry points on the bench model, under tuning-knob settings.
usage: python tools/diag/fused_calls.py [knob=value ...]   knobs: lds_cap, tiles, waves, groups, wg_per_cu"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pdm_ssd_amd import _native, fused
dev = torch.device("cuda:0")
backbone, neck = bench.build_models(dev)
_, points = bench.make_batch(32, 16384, "uniform", 1234, dev)
lib = _native.lib()

def run():
    return backbone({'batch_size': 32, 'points': points, 'points_per_sample_checked': True})

def measure(reps=3):
    with torch.no_grad():
        run(); torch.cuda.synchronize()
        with bench.OpTimer() as t:
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            recs = [(n, e0.elapsed_time(e1)) for n, ints, e0, e1 in t.records if "mlp_fused" in n or "mlp_gemm" in n or "mlp_packed" in n]
    per = len(recs) // reps
    out = []
    for i in range(per):
        out.append((recs[i][0], sum(recs[i + k * per][1] for k in range(reps)) / reps))
    return out

settings = [dict(kv.split("=") for kv in arg.split(",") if kv != "default") for arg in sys.argv[1:]] or [{}]
cols = []
for st in settings:
    for k, v in st.items():
        getattr(lib, "pdm_tune_fused_" + k)(int(v))
    cols.append(measure())
    for k in st:
        getattr(lib, "pdm_tune_fused_" + k)(0 if k != "lds_cap" else 152 * 1024)
print("%-24s" % "call" + "".join("%22s" % ",".join("%s=%s" % kv for kv in st.items()) for st in settings))
for i, (name, _) in enumerate(cols[0]):
    print("%-24s" % name.replace("pdm_", "") + "".join("%22.1f" % (c[i][1] * 1e3) for c in cols))
print("%-24s" % "total us" + "".join("%22.1f" % (sum(x[1] for x in c) * 1e3) for c in cols))
