"""What does the vendor fp32 GEMM reach on the wide-layer shapes?  (ceiling check, not used by the product)"""
import torch
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [("FP1 l1", 524288, 272, 128), ("FP1 l2", 524288, 128, 128), ("FP2 l1", 131072, 608, 256), ("FP3 l1", 32768, 768, 512),
          ("FP4 l1", 8192, 1536, 512), ("SA3s1 l1", 262144, 259, 128), ("SA3s1 l3", 262144, 196, 256), ("SA4s1 l1", 65536, 515, 256),
          ("SA4s1 l3", 65536, 384, 512), ("SA2s1 l3", 1048576, 96, 128)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev)
    ms = t(lambda: a @ b)
    print(f"{name:10s} M={M:8d} K={K:5d} N={N:4d}: {ms:7.3f} ms  {2*M*K*N/ms/1e9:7.1f} TFLOP/s")
