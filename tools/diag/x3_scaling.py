"""Time per row of pdm_rows_mlp_x3 (one stack, 128 -> 256 -> 256 -> 8) against the number of rows and the grid cap: how much of a
launch is ramp-up / tail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native, fused
from pdm_ssd_amd.dense_heads.point_head_box import _fc_layers
from pdm_ssd_amd.dense_heads.point_head_template import PointHeadTemplate
if "--lib" in sys.argv:
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]); print("library:", _native.LIB_PATH)
dev = torch.device("cuda:0"); l = _native.lib()
torch.manual_seed(0)
seq = PointHeadTemplate.make_fc_layers([256, 256], 128, 8).eval()
layers = _fc_layers(seq)
px3, p32 = fused.PackedMLPx3(layers, dev), fused.PackedMLP(layers, dev)
def timed(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
big = torch.randn(4 * 524288, 128, device=dev)
out = torch.empty(4 * 524288, 8, device=dev)
for _ in range(30): fused.rows_forward(p32, big[:524288], out[:524288], relu_last=False)
for rows in (65536, 131072, 262144, 524288, 1048576, 2097152):
    x, o = big[:rows], out[:rows]
    line = f"rows {rows:8d}:"
    for wg in (2, 4, 12):
        old = l.pdm_tune_rows_x3_wg_per_cu(wg)
        ms = timed(lambda: fused.rows_forward_x3(px3, x, o, relu_last=False))
        l.pdm_tune_rows_x3_wg_per_cu(old)
        line += f"  x3 cap {wg:2d}/CU {ms * 1e3:8.1f} us ({ms * 1e6 / rows:6.3f} ns/row, {rows * px3.flops_per_position / ms / 1e9:6.1f} TF)"
    ms = timed(lambda: fused.rows_forward(p32, x, o, relu_last=False))
    line += f" | fp32 {ms * 1e3:8.1f} us ({rows * p32.flops_per_position / ms / 1e9:6.1f} TF)"
    print(line, flush=True)
