#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
dev = "cuda"
def bench(M, N, Kd, cdt=torch.bfloat16, rounds=30):
    A = torch.randn(M, Kd).to(torch.bfloat16).to(dev)
    B = torch.randn(N, Kd).to(torch.bfloat16).to(dev)
    C = torch.zeros(M, N, dtype=cdt, device=dev)
    kw = dict(M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N)
    for _ in range(3): K.gemm(A, B, C, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds): K.gemm(A, B, C, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / rounds * 1e3
    print("%s M%6d N%6d K%6d : %7.1f us  %6.1f TF/s" % (os.environ.get("S2T_HIP_LIB", "default")[-20:], M, N, Kd, us, 2.0 * M * N * Kd / us / 1e6), flush=True)
for Kd in (64, 256, 1024):
    bench(16000, 2048, Kd)
bench(16000, 256, 2048)
