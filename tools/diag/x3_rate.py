"""fp32 emulated on the bf16 matrix pipe (pdm_rows_mlp_x3) against the fp32-MFMA chain kernel on the point head's shapes
(bs = 32 x 16384 rows, 128 -> 256 -> 256 -> {3, 8}): time, equivalent fp32 TFLOP/s, error table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native, fused
from pdm_ssd_amd.dense_heads.point_head_box import _fc_layers
from pdm_ssd_amd.dense_heads.point_head_template import PointHeadTemplate
if "--lib" in sys.argv:
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    print("library:", _native.LIB_PATH, flush=True)
dev = torch.device("cuda:0"); l = _native.lib()
torch.manual_seed(0)
rows = 32 * 16384
def timed(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(rows, 128, device=dev)
packs = []
for cout in (3, 8):
    seq = PointHeadTemplate.make_fc_layers([256, 256], 128, cout).eval()
    for m in seq.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    layers = _fc_layers(seq)
    packs.append((cout, seq, fused.PackedMLP(layers, dev), fused.PackedMLPx3(layers, dev)))
outs32 = [torch.empty(rows, (c + 3) // 4 * 4, device=dev) for c, *_ in packs]
outsx3 = [torch.empty(rows, (c + 3) // 4 * 4, device=dev) for c, *_ in packs]
def f32_pair(): fused.rows_forward_pair(packs[0][2], packs[1][2], x, outs32[0], outs32[1], relu_last=False)
def x3_two():
    for (c, s, p32, px3), o in zip(packs, outsx3): fused.rows_forward_x3(px3, x, o, relu_last=False)
for _ in range(60): f32_pair()      # the clock settles over the first ~100 ms of load
flop = rows * sum(p[2].flops_per_position for p in packs)
for rep in range(3):
    a, b = timed(f32_pair), timed(x3_two)
    print(f"fp32 MFMA (one launch, both stacks) {a:.4f} ms = {flop / a / 1e9:.1f} TFLOP/s | 3 x bf16 emulation (two launches) {b:.4f} ms = "
          f"{flop / b / 1e9:.1f} TFLOP/s fp32-equivalent | speed-up {a / b:.2f}x", flush=True)
for n in (2, 4, 8, 12, 16):
    old = l.pdm_tune_rows_x3_wg_per_cu(n)
    print(f"x3, {n:2d} workgroups per CU: {timed(x3_two, 5):.4f} ms", flush=True)
    l.pdm_tune_rows_x3_wg_per_cu(old)
# error table on the first 65536 rows against float64 on the CPU
sub = 65536
for (c, seq, p32, px3), o32, ox3 in zip(packs, outs32, outsx3):
    with torch.no_grad():
        want = seq.double()(x[:sub].cpu().double())
    a, b = o32[:sub, :c].cpu().double(), ox3[:sub, :c].cpu().double()
    scale = float(want.abs().max())
    print(f"cout={c}: output scale {scale:.3f} | max |x3 - fp32 MFMA| / scale {float((a - b).abs().max()) / scale:.3e} | fp32 MFMA vs float64 "
          f"{float((a - want).abs().max()) / scale:.3e} (rms {float((a - want).pow(2).mean().sqrt()) / scale:.3e}) | x3 vs float64 "
          f"{float((b - want).abs().max()) / scale:.3e} (rms {float((b - want).pow(2).mean().sqrt()) / scale:.3e})", flush=True)
