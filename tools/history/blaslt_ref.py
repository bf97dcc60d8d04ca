#!/usr/bin/env python3
"""Calibration only: what the vendor library (torch.matmul -> hipBLASLt) reaches on the hot-path GEMM shapes."""
import torch
dev = "cuda"
def bench(M, N, Kd, rounds=30):
    A = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    B = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(A, B.t(), out=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds): torch.matmul(A, B.t(), out=C)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / rounds * 1e3
    print("M%6d N%6d K%6d : %7.1f us  %6.1f TF/s" % (M, N, Kd, us, 2.0 * M * N * Kd / us / 1e6), flush=True)
for s in [(16000, 2048, 256), (16000, 256, 2048), (16000, 256, 256), (16000, 768, 256), (16000, 10000, 256), (16000, 3072, 256), (3904, 10000, 256),
          (16000, 256, 10000), (3904, 256, 10000), (16000, 2048, 1024), (8192, 8192, 8192)]:
    bench(*s)
