#!/usr/bin/env python3
"""Micro-benchmark of s2t_gemm on chosen shapes (HIP events, interleaved rounds in one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

dev = "cuda"
def bench(M, N, Kd, epi="none", akm=False, bkm=False, cdt=torch.bfloat16, split=1, rounds=20):
    g = torch.Generator(device="cpu").manual_seed(0)
    A = torch.randn((Kd, M) if akm else (M, Kd), generator=g).to(torch.bfloat16).to(dev)
    B = torch.randn((Kd, N) if bkm else (N, Kd), generator=g).to(torch.bfloat16).to(dev)
    C = torch.zeros(M, N, dtype=cdt, device=dev)
    kw = dict(M=M, N=N, K=Kd, lda=M if akm else Kd, ldb=N if bkm else Kd, ldc=N, a_kmajor=akm, b_kmajor=bkm)
    if epi == "ffn1":
        kw.update(bias=torch.zeros(N, device=dev), act="swish", preact=torch.empty_like(C), ldp=N)
    elif epi == "bias":
        kw.update(bias=torch.zeros(N, device=dev))
    elif epi == "res":
        kw.update(bias=torch.zeros(N, device=dev), residual=torch.zeros_like(C), ldr=N, alpha=0.5)
    elif epi == "atomic":
        kw.update(split_k=split, c_atomic=True)
    for _ in range(3):
        K.gemm(A, B, C, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        K.gemm(A, B, C, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / rounds * 1e3
    print("M%6d N%6d K%6d %-7s %s%s c=%s split=%d : %8.1f us  %7.1f TF/s" % (M, N, Kd, epi, "AK" if akm else "AR", "BK" if bkm else "BR",
          "f32" if cdt == torch.float32 else "bf16", split, us, 2.0 * M * N * Kd / us / 1e6), flush=True)

print("mode:", "ring" if os.environ.get("S2T_GEMM_RING") == "1" else "generic")
bench(8192, 8192, 8192)
bench(4096, 4096, 4096)
bench(16000, 2048, 256, "none")
bench(16000, 2048, 256, "bias")
bench(16000, 2048, 256, "ffn1")
bench(16000, 256, 2048, "none")
bench(16000, 256, 2048, "res")
bench(16000, 768, 256, "bias")
bench(16000, 256, 256, "res")
bench(16000, 2048, 256, "none", bkm=True)
bench(2048, 256, 16000, "atomic", akm=True, bkm=True, cdt=torch.float32, split=8)
bench(256, 256, 16000, "atomic", akm=True, bkm=True, cdt=torch.float32, split=64)
