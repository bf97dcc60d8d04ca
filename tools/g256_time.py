#!/usr/bin/env python3
"""Times of s2t_gemm's large-tile path on a few shapes with the library S2T_HIP_LIB names (experiment builds: tools/g256_dbg.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
dev = torch.device("cuda", 0)
torch.manual_seed(0)
K.gemm_configure(2)
res = []
for M, N, Kd in [(64000, 2048, 512), (64000, 512, 2048), (64000, 512, 512), (16000, 10000, 256), (8192, 8192, 8192)]:
    A = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts = []
    for r in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            K.gemm(A, B, C, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=b)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 250)
    t = sorted(ts)[4]
    res.append("%dx%dx%d %.1f us %.0f TF" % (M, N, Kd, t, 2.0 * M * N * Kd / t / 1e6))
print((os.environ.get("S2T_HIP_LIB") or "x/shipped/x").split("/")[-2], " | ".join(res), flush=True)
