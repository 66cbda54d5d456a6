"""Heat-map head in one launch (pdm_bev_head_fused, rows_chain_kernel<8,4,4,1,true>) at the bench shape: patches dealt to
the XCDs in contiguous ranges against launch order, settled clock.  Under rocprofv3 --pmc the same script gives the HBM
traffic of each form (tools/pmc_traffic.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native
if "--lib" in sys.argv:      # an alternative build of the library (e.g. one compiled with -DDW_SL=2)
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    print("library:", _native.LIB_PATH, flush=True)
from pdm_ssd_amd.detector_config import build_pdm_ssd
dev = torch.device("cuda:0")
m = build_pdm_ssd().to(dev).eval()
l = _native.lib()
B = 32
sf = torch.randn(B, 188, 188, 128, device=dev).permute(0, 3, 1, 2)
with torch.no_grad():
    for _ in range(150): m.dense_head({'spatial_features': sf})
    torch.cuda.synchronize()
    for on in (1, 0, 1, 0):
        l.pdm_tune_rows_chain_xcd(on)
        m.dense_head({'spatial_features': sf}); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m.dense_head({'spatial_features': sf})
        e1.record(); torch.cuda.synchronize()
        print(f"heat-map head, patches by XCD range = {on}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
l.pdm_tune_rows_chain_xcd(1)
with torch.no_grad():
    for n in (2, 3, 4, 6, 8, 12, 24):        # grid cap = 256 CUs x n workgroups (two resident per CU)
        old = l.pdm_tune_rows_chain_dw_wg_per_cu(n)
        m.dense_head({'spatial_features': sf}); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m.dense_head({'spatial_features': sf})
        e1.record(); torch.cuda.synchronize()
        print(f"heat-map head, grid = 256 x {n:2d} workgroups: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
        l.pdm_tune_rows_chain_dw_wg_per_cu(old)
