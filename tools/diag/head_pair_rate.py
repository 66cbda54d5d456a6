"""The point head's two stacks: ONE launch (pdm_rows_mlp_fused_pair, rows_chain_pair_kernel) against two launches of
rows_chain_kernel, settled clock, bs = 32 x 16384 rows; also the grid cap (workgroups per CU) for the pair kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native, fused
from pdm_ssd_amd.dense_heads.point_head_box import _fc_layers
from pdm_ssd_amd.dense_heads.point_head_template import PointHeadTemplate
dev = torch.device("cuda:0"); l = _native.lib()
torch.manual_seed(0)
rows = 32 * 16384
sa = PointHeadTemplate.make_fc_layers([256, 256], 128, 3).to(dev).eval()
sb = PointHeadTemplate.make_fc_layers([256, 256], 128, 8).to(dev).eval()
pa, pb = fused.PackedMLP(_fc_layers(sa), dev), fused.PackedMLP(_fc_layers(sb), dev)
x = torch.randn(rows, 128, device=dev)
oa, ob = torch.empty(rows, 4, device=dev), torch.empty(rows, 8, device=dev)
flop = rows * (pa.flops_per_position + pb.flops_per_position)

def two():
    fused.rows_forward(pa, x, oa, relu_last=False); fused.rows_forward(pb, x, ob, relu_last=False)
def one():
    fused.rows_forward_pair(pa, pb, x, oa, ob, relu_last=False)
def timed(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(80): two()     # the clock settles over the first ~100 ms of load
two(); ra, rb = oa.clone(), ob.clone()
one(); torch.cuda.synchronize()
print("bit-identical:", bool(torch.equal(ra, oa) and torch.equal(rb, ob)), flush=True)
for rep in range(3):
    t2, t1 = timed(two), timed(one)
    print(f"two launches {t2:.4f} ms ({flop / t2 / 1e9:.1f} TFLOP/s)   one launch {t1:.4f} ms ({flop / t1 / 1e9:.1f} TFLOP/s)", flush=True)
for n in (12, 2, 3, 4, 6, 8, 16, 12):
    old = l.pdm_tune_rows_chain_wg_per_cu(n)
    print(f"pair kernel, {n:2d} workgroups per CU: {timed(one, 5):.4f} ms", flush=True)
    l.pdm_tune_rows_chain_wg_per_cu(old)
