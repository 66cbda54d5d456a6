"""pdm_rows_mlp_fused on the hybrid head's shapes: register-resident chain (rows_chain.hip) vs the general chain kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native, fused
from pdm_ssd_amd.dense_heads.point_head_box import _fc_layers
from pdm_ssd_amd.dense_heads.point_head_template import PointHeadTemplate
if len(sys.argv) > 2 and sys.argv[1] == "--lib":    # a timing build from `make -C pdm_ssd_amd/csrc diag-rc`
    _native.LIB_PATH = os.path.abspath(sys.argv[2])
    print("library:", _native.LIB_PATH, flush=True)
dev = torch.device("cuda:0"); l = _native.lib()
torch.manual_seed(0)
for name, cin, fc, cout, rows in [("point head cls", 128, [256, 256], 3, 32 * 16384), ("point head box", 128, [256, 256], 8, 32 * 16384),
                                   ("heat-map cells", 128, [64, 64], 3, 32 * 200 * 176)]:
    seq = PointHeadTemplate.make_fc_layers(fc, cin, cout).to(dev).eval()
    pk = fused.PackedMLP(_fc_layers(seq), dev)
    x = torch.randn(rows, cin, device=dev)
    out = torch.empty(rows, (cout + 3) // 4 * 4, device=dev)
    ref = None
    for _ in range(150): fused.rows_forward(pk, x, out, relu_last=False)   # the clock settles over the first ~100 ms of load
    for chain in ((1,) if len(sys.argv) > 2 else (1, 0)):
        l.pdm_tune_fused_chain(chain)
        fused.rows_forward(pk, x, out, relu_last=False); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fused.rows_forward(pk, x, out, relu_last=False)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        if ref is None: ref = out.clone()
        print(f"{name:16s} rows={rows} chain={chain}: {ms:.3f} ms  {rows * pk.flops_per_position / ms / 1e9:.1f} TFLOP/s  max|diff|={float((out - ref).abs().max()):.2e}", flush=True)
l.pdm_tune_fused_chain(1)
if len(sys.argv) > 2: sys.exit(0)
# grid cap of the chain kernel (workgroups per CU; two are resident): a workgroup's FIRST tile is slow (its rows are not
# prefetched, tools/diag/rows_chain_phase.py), so fewer, longer-lived workgroups help until the tail grows
seq = PointHeadTemplate.make_fc_layers([256, 256], 128, 3).to(dev).eval()
pk = fused.PackedMLP(_fc_layers(seq), dev)
rows = 32 * 16384
x = torch.randn(rows, 128, device=dev); out = torch.empty(rows, 4, device=dev)
for _ in range(200): fused.rows_forward(pk, x, out, relu_last=False)   # the clock settles over the first ~100 ms of load
torch.cuda.synchronize()
for n in (12, 2, 3, 4, 6, 8, 12, 16, 2, 4, 12):
    old = l.pdm_tune_rows_chain_wg_per_cu(n)
    fused.rows_forward(pk, x, out, relu_last=False); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fused.rows_forward(pk, x, out, relu_last=False)
    e1.record(); torch.cuda.synchronize()
    print(f"point head cls, {n:2d} workgroups per CU: {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
    l.pdm_tune_rows_chain_wg_per_cu(old)
