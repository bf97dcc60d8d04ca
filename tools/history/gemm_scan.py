#!/usr/bin/env python3
"""K / N scans of s2t_gemm at M=16000 to separate the per-tile fixed cost (prologue + epilogue) from the K-loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
dev = "cuda"
def bench(M, N, Kd, cdt=torch.bfloat16, rounds=30, bkm=False):
    A = torch.randn(M, Kd).to(torch.bfloat16).to(dev)
    B = (torch.randn(Kd, N) if bkm else torch.randn(N, Kd)).to(torch.bfloat16).to(dev)
    C = torch.zeros(M, N, dtype=cdt, device=dev)
    kw = dict(M=M, N=N, K=Kd, lda=Kd, ldb=N if bkm else Kd, ldc=N, b_kmajor=bkm)
    for _ in range(3): K.gemm(A, B, C, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds): K.gemm(A, B, C, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / rounds * 1e3
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print("M%6d N%6d K%6d c=%s %s tiles %5d : %7.1f us  %6.1f TF/s  C-write %5.2f TB/s" % (M, N, Kd, "f32" if cdt == torch.float32 else "bf16", "BK" if bkm else "BR", tiles, us, 2.0 * M * N * Kd / us / 1e6, M * N * C.element_size() / us / 1e6), flush=True)
for Kd in (64, 128, 256, 512, 1024):
    bench(16000, 2048, Kd)
bench(16000, 2048, 256, torch.float32)
for N in (256, 512, 1024, 4096):
    bench(16000, N, 256)
for M in (2048, 4096, 8192, 32768):
    bench(M, 2048, 256)
