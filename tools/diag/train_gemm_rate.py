"""pdm_tg_gemm_nt / pdm_tg_wgrad on the layer shapes of the bs = 32 training step: per-launch time (HIP events over back-to-back
launches), algorithmic GB/s (X + Y once; dY + X once) and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import train_gemm as tg
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("SA1s1 L1", 4194304, 8, 32), ("SA1s1 L2", 4194304, 32, 32), ("SA1s1 L3", 4194304, 32, 64), ("SA1s0 L3", 2097152, 16, 32),
          ("SA2s1 L1", 1048576, 104, 64), ("SA2s1 L2", 1048576, 64, 96), ("SA2s1 L3", 1048576, 96, 128),
          ("SA3s1 L1", 262144, 264, 128), ("SA3s1 L2", 262144, 128, 200), ("SA3s1 L3", 262144, 200, 256),
          ("SA4s1 L1", 65536, 520, 256), ("SA4s1 L3", 65536, 384, 512),
          ("FP1 L1", 524288, 264, 128), ("FP1 L2", 524288, 128, 128), ("FP2 L1", 131072, 608, 256), ("FP3 L1", 32768, 768, 512),
          ("head L1", 524288, 128, 256), ("head L2", 524288, 256, 256), ("head L3", 524288, 256, 8), ("hm L1", 1126400, 128, 64), ("hm L2", 1126400, 64, 64)]
tot = [0.0, 0.0, 0.0]
for name, R, K, N in shapes:
    x = torch.randn(R, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    dy = torch.randn(R, N, device=dev).bfloat16()
    wt = w.t().contiguous()
    y = torch.empty(R, N, dtype=torch.bfloat16, device=dev)
    dx = torch.empty(R, K, dtype=torch.bfloat16, device=dev)
    f = t(lambda: tg.gemm_nt(x, w, stats=True, out=y))
    g = t(lambda: tg.gemm_nt(dy, wt, out=dx))
    h = t(lambda: tg.wgrad(dy, x))
    by = 2.0 * R * (K + N)
    fl = 2.0 * R * K * N
    tot[0] += f; tot[1] += g; tot[2] += h
    print(f"{name:9s} R={R:8d} K={K:4d} N={N:4d}: fwd {f:7.1f} us {by/1e3/f:6.0f} GB/s {fl/1e6/f:6.1f} TF | dgrad {g:7.1f} us {by/1e3/g:6.0f} GB/s | wgrad {h:7.1f} us {by/1e3/h:6.0f} GB/s", flush=True)
print("sum us: fwd %.0f dgrad %.0f wgrad %.0f" % tuple(tot))
