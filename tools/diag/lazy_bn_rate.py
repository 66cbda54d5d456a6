"""BatchNorm + ReLU backward inside the data gradient (pdm_tg_gemm_nt_dy) against the two separate launches it replaces
(pdm_bn_relu_backward_apply, then pdm_tg_gemm_nt on the formed gradient), per layer shape of the bs = 32 training step:
R rows, K = channels of the BatchNorm (the contraction), N = width of the data gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import _native, train_gemm as tg
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("SA1s1 L2", 4194304, 32, 8), ("SA1s1 L3", 4194304, 32, 32), ("SA1s0 L2", 2097152, 16, 8), ("SA1s0 L3", 2097152, 16, 16),
          ("SA2s1 L2", 1048576, 64, 104), ("SA2s1 L3", 1048576, 96, 64), ("SA3s1 L2", 262144, 128, 264), ("SA3s1 L3", 262144, 200, 128),
          ("SA4s1 L3", 65536, 384, 256), ("FP1 L2", 524288, 128, 264), ("FP2 L2", 131072, 256, 608), ("head L2", 524288, 256, 128),
          ("head L3", 524288, 256, 256), ("hm L2", 1126400, 64, 128), ("hm L3", 1126400, 64, 64)]
s = torch.cuda.current_stream().cuda_stream
for name, R, K, N in shapes:
    dz = torch.randn(R, K, device=dev).bfloat16()
    y = torch.randn(R, K, device=dev).bfloat16()
    coef = torch.rand(4, K, device=dev) + 0.5
    grads = torch.randn(4, K, device=dev) * 0.01
    wt = torch.randn(N, K, device=dev).bfloat16()
    dy = torch.empty_like(y)
    a = t(lambda: _native.call("pdm_bn_relu_backward_apply", s, 1, 0, R, K, 1, y.data_ptr(), dz.data_ptr(), dy.data_ptr(), coef.data_ptr(), grads.data_ptr(), 1))
    g = t(lambda: tg.gemm_nt(dy, wt))
    f = t(lambda: tg.gemm_nt_dy(dz, y, coef, grads, wt))
    print(f"{name:9s} R={R:8d} K={K:4d} N={N:4d}: apply {a:7.1f} + dgrad {g:7.1f} = {a + g:7.1f} us | fused {f:7.1f} us  ({(a + g) / f:4.2f}x)", flush=True)
