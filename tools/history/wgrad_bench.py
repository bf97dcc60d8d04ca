#!/usr/bin/env python3
"""Weight-gradient GEMMs (TN, split-K) of the model: two-phase workspace reduction vs float atomics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
dev = "cuda"
def run(Nout, Kin, M, split, ws, rounds=30):
    dY = torch.randn(M, Nout).to(torch.bfloat16).to(dev); X = torch.randn(M, Kin).to(torch.bfloat16).to(dev)
    dW = torch.zeros(Nout, Kin, device=dev); db = torch.zeros(Nout, device=dev)
    f = lambda: K.gemm(dY, X, dW, M=Nout, N=Kin, K=M, lda=Nout, ldb=Kin, ldc=Kin, a_kmajor=True, b_kmajor=True,
                       split_k=split, c_atomic=True, colsum_a=db, splitk_workspace=ws)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rounds * 1e3
for (Nout, Kin, M) in [(2048, 256, 16000), (256, 2048, 16000), (256, 256, 16000), (512, 256, 16000), (768, 256, 16000),
                       (256, 256, 3904), (2048, 256, 3904), (10000, 256, 16000)]:
    tiles = ((Nout + 127) // 128) * ((Kin + 127) // 128)
    ktiles = (M + 63) // 64
    for split in sorted({max(1, min((ktiles + 3) // 4, (512 + tiles - 1) // tiles)), max(1, min((ktiles + 7) // 8, (256 + tiles - 1) // tiles)),
                         max(1, min((ktiles + 1) // 2, (1024 + tiles - 1) // tiles))}):
        a = run(Nout, Kin, M, split, True); b = run(Nout, Kin, M, split, False)
        print("dW %5dx%5d M%6d tiles %3d split %3d : workspace %6.1f us   atomics %6.1f us   (%.0f TF/s best)" % (
            Nout, Kin, M, tiles, split, a, b, 2.0 * Nout * Kin * M / min(a, b) / 1e6), flush=True)
