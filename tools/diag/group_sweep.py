"""group_points (LDS rows form) parameter sweep on the six large launches of the API-exact section: rows per workgroup x
parts of L x threads, against round 2's kernel (variant 2) — per-launch HIP-event times, uniform-cloud ball-query indices
(what bench.py uses) and random indices (worst case for the LDS gather)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native
from pdm_ssd_amd.pointnet2_batch import pointnet2_utils as pu
dev = torch.device("cuda:0")
lib = _native.lib()
B = 32
def t(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def pack(variant, rpw=0, ls=0, th=1, uq=0):
    return variant | rpw << 4 | ls << 8 | th << 16 | uq << 20
shapes = [(96, 4096, 1024, 16), (96, 4096, 1024, 32), (256, 1024, 256, 16), (256, 1024, 256, 32), (512, 256, 64, 16), (512, 256, 64, 32)]
tot = {}
for (C, N, M, ns) in shapes:
    f = torch.randn(B, C, N, device=dev)
    for kind in ("padded", "random"):
        if kind == "padded":   # a ball that holds only its centre: the first hit fills every slot
            idx = torch.randint(0, N, (B, M, 1), dtype=torch.int32, device=dev).expand(B, M, ns).contiguous()
        else:
            idx = torch.randint(0, N, (B, M, ns), dtype=torch.int32, device=dev)
        by = B * (4 * M * ns + 4 * C * N + 4 * C * M * ns)
        ref = None
        res = []
        cands = [("r2", pack(2))]
        for th in (1, 2, 4):
            for rpw in (1, 2, 3, 4, 6, 8):
                if rpw * N * 4 > 65536 or rpw > C: continue
                for ls in (1, 2, 4, 8, 16):
                    if M * ns // 4 // ls < 256 * th: continue
                    cands.append((f"T{256*th} rpw{rpw} ls{ls}", pack(1, rpw, ls, th)))
        for v, u, tag in ((3, 0, "noxcd"), (1, 1, "uq1"), (1, 2, "uq2")):     # what in the rows kernel matters
            for th, rpw, ls in ((1, min(8, 32768 // (N * 4)), 1), (4, min(8, 65536 // (N * 4)), 2), (2, min(8, 49152 // (N * 4)), 1)):
                if rpw >= 1 and M * ns // 4 // ls >= 256 * th:
                    cands.append((f"{tag} T{256*th} rpw{rpw} ls{ls}", pack(v, rpw, ls, th, u)))
        cands.append(("auto", 0))
        for name, p in cands:
            lib.pdm_tune_group_rows(p)
            out = pu.grouping_operation(f, idx)
            if ref is None: ref = out
            else: assert torch.equal(out, ref), name
            us = t(lambda: pu.grouping_operation(f, idx))
            res.append((us, name))
        lib.pdm_tune_group_rows(0)
        r2 = [u for u, nme in res if nme == "r2"][0]
        au = [u for u, nme in res if nme == "auto"][0]
        best = sorted(res)[:6]
        print(f"C={C} N={N} M={M} ns={ns} {kind}: {by/1e6:.0f} MB  r2 {r2:.1f} us ({by/1e3/r2:.0f} GB/s)  auto {au:.1f} us ({by/1e3/au:.0f} GB/s)  best: " +
              ", ".join(f"{nme} {u:.1f} ({by/1e3/u:.0f})" for u, nme in best), flush=True)
        tot.setdefault(kind, [0, 0, 0]); tot[kind][0] += r2; tot[kind][1] += au; tot[kind][2] += best[0][0]
print({k: [round(x, 1) for x in v] for k, v in tot.items()}, "(sum us: r2, auto, best)")
