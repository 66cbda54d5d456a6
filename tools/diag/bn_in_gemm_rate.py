"""BatchNorm + ReLU inside the NEXT contraction's load path (pdm_tg_gemm_nt / pdm_tg_wgrad with x_bn_coef) against the separate
operator (pdm_bn_relu_forward_stats' apply pass, then the plain contractions), per layer shape of the bs = 32 training step:
R rows, K = channels of the BatchNorm (= the next layer's input), N = the next layer's output width."""
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
shapes = [("SA1s1 L2", 4194304, 32, 32), ("SA1s1 L3", 4194304, 32, 64), ("SA1s0 L2", 2097152, 16, 16), ("SA1s0 L3", 2097152, 16, 32),
          ("SA2s1 L2", 1048576, 64, 96), ("SA2s1 L3", 1048576, 96, 128), ("SA3s1 L2", 262144, 128, 200), ("SA3s1 L3", 262144, 200, 256),
          ("SA4s1 L3", 65536, 384, 512), ("FP1 L2", 524288, 128, 128), ("FP2 L2", 131072, 256, 256), ("head L2", 524288, 256, 256),
          ("head L3", 524288, 256, 8), ("hm L2", 1126400, 64, 64), ("hm L3", 1126400, 64, 8)]
s = torch.cuda.current_stream().cuda_stream
for name, R, K, N in shapes:
    y = torch.randn(R, K, device=dev).bfloat16()
    z = torch.empty_like(y)
    coef = torch.rand(4, K, device=dev) + 0.5
    w = torch.randn(N, K, device=dev).bfloat16()
    dyn = torch.randn(R, N, device=dev).bfloat16()
    gamma = torch.ones(K, device=dev); beta = torch.zeros(K, device=dev); rm = torch.zeros(K, device=dev); rv = torch.ones(K, device=dev)
    _, st = tg.gemm_nt(torch.randn(R, 8, device=dev).bfloat16(), torch.randn(K, 8, device=dev).bfloat16(), stats=True)
    c2 = torch.empty(4, K, device=dev)
    a = t(lambda: _native.call("pdm_bn_relu_forward_stats", s, 1, R, K, y.data_ptr(), z.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1,
                               rm.data_ptr(), rv.data_ptr(), c2.data_ptr(), st.data_ptr(), st.shape[0], 1))
    f0 = t(lambda: tg.gemm_nt(z, w)); f1 = t(lambda: tg.gemm_nt(y, w, x_bn_coef=coef))
    g0 = t(lambda: tg.wgrad(dyn, z)); g1 = t(lambda: tg.wgrad(dyn, y, x_bn_coef=coef))
    print(f"{name:9s} R={R:8d} K={K:4d} N={N:4d}: apply {a:6.1f} + fwd {f0:6.1f} + wgrad {g0:6.1f} = {a + f0 + g0:7.1f} us | in load path: fwd {f1:6.1f} + wgrad {g1:6.1f} = {f1 + g1:7.1f} us  ({(a + f0 + g0) / (f1 + g1):4.2f}x)", flush=True)
