#!/usr/bin/env python3
"""How the vendor GEMM treats the tall-skinny products of the train step (rows x K) @ (K x N), bf16: the plain call,
the same rows cut into slabs and batched, and the transposed product.  HIP-event timing, GB/s of the minimal traffic.
    python tools/diag/tall_gemm_forms.py
"""
import torch

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rows, K, N in ((524288, 128, 256), (524288, 256, 256), (2097152, 32, 64), (1048576, 64, 128), (524288, 128, 128)):
    x = torch.randn(rows, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()
    mb = (rows * K + rows * N) * 2 / 1e6
    forms = {
        "x @ w.t()": lambda: x @ w.t(),
        "x @ wt (K x N contiguous)": lambda: x @ wt,
        "linear": lambda: torch.nn.functional.linear(x, w),
        "bmm 64 slabs": lambda: torch.bmm(x.view(64, rows // 64, K), wt.expand(64, K, N)),
        "bmm 512 slabs": lambda: torch.bmm(x.view(512, rows // 512, K), wt.expand(512, K, N)),
        "(w @ x.t()).t()": lambda: (w @ x.t()).t(),
    }
    print(f"rows={rows} K={K} N={N}: minimal traffic {mb:.0f} MB")
    for name, fn in forms.items():
        us = timeit(fn)
        print(f"   {name:28s} {us:8.1f} us  {mb / us * 1e-3 * 1e3:7.0f} GB/s  {2.0 * rows * K * N / us * 1e-6:6.0f} TFLOP/s", flush=True)

print("input gradient of the rows x K product: dx (rows x K) = dy (rows x N) @ w (N x K)")
for rows, K, N in ((524288, 128, 256), (2097152, 32, 64), (1048576, 64, 128)):
    dy = torch.randn(rows, N, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()                      # (K, N)
    mb = (rows * K + rows * N) * 2 / 1e6
    for name, fn in {"dy @ w": lambda: dy @ w, "dy @ wt.t() (wt = w.t().contiguous())": lambda: dy @ wt.t(),
                     "(wt @ dy.t()).t()": lambda: (wt @ dy.t()).t()}.items():
        us = timeit(fn)
        print(f"   rows={rows} K={K} N={N} {name:40s} {us:8.1f} us  {mb / us:7.2f} TB/s", flush=True)

print("position-fastest tensors (B, C, L): y_b = w (N x K) @ x_b (K x L)")
for B, K, N, L in ((32, 257, 128, 16384), (32, 256, 256, 4096), (32, 768, 512, 1024)):
    x = torch.randn(B, K, L, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    mb = (B * K * L + B * N * L) * 2 / 1e6
    for name, fn in {"matmul(w, x)": lambda: torch.matmul(w, x),
                     "matmul(x^T, w^T)^T": lambda: torch.matmul(x.transpose(1, 2), w.t()).transpose(1, 2),
                     "bmm(w expanded, x)": lambda: torch.bmm(w.expand(B, N, K), x)}.items():
        us = timeit(fn)
        print(f"   B={B} K={K} N={N} L={L} {name:24s} {us:8.1f} us  {mb / us:7.2f} TB/s", flush=True)
