"""A few layer shapes of the training contractions, 5 launches each, for a rocprofv3 --pmc pass (SQ counters per kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pdm_ssd_amd import train_gemm as tg
dev = torch.device("cuda:0")
for name, R, K, N in [("SA1s1 L2", 4194304, 32, 32), ("SA2s1 L3", 1048576, 96, 128), ("head L2", 524288, 256, 256), ("hm L1", 1126400, 128, 64)]:
    x = torch.randn(R, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); dy = torch.randn(R, N, device=dev).bfloat16()
    y = torch.empty(R, N, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        tg.gemm_nt(x, w, stats=True, out=y)
        tg.wgrad(dy, x)
    torch.cuda.synchronize()
