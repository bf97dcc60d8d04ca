#!/usr/bin/env python3
"""One GEMM shape in a loop, for rocprofv3 --pmc passes (tools/pmc_gemm.sh): kind = fwd | wgrad."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
kind = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
dev = "cuda"
if kind == "wgrad":
    M, Nout, Kin = 16000, 2048, 2048   # many tiles, K = 16000: the K-loop of the k-major kernel without split-K effects
    dY = torch.randn(M, Nout).to(torch.bfloat16).to(dev); X = torch.randn(M, Kin).to(torch.bfloat16).to(dev)
    dW = torch.zeros(Nout, Kin, device=dev)
    f = lambda: K.gemm(dY, X, dW, M=Nout, N=Kin, K=M, lda=Nout, ldb=Kin, ldc=Kin, a_kmajor=True, b_kmajor=True)
    flops = 2.0 * M * Nout * Kin
else:
    M, N, Kd = 16384, 2048, 4096
    A = torch.randn(M, Kd).to(torch.bfloat16).to(dev); B = torch.randn(N, Kd).to(torch.bfloat16).to(dev)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda: K.gemm(A, B, C, M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N)
    flops = 2.0 * M * N * Kd
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
print("%s: %.1f us  %.0f TF/s" % (kind, us, flops / us / 1e6))
