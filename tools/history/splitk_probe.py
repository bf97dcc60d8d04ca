#!/usr/bin/env python3
"""Two-phase split-K products (input gradients with a long K over few output tiles) on the 128 x 128 kernel against the large
tiles of gemm256.hip (work item = (tile, split)); us per call (both phases), same split on both sides."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
dev = "cuda"
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(16000, 256, 3072), (13142, 256, 3072), (16000, 256, 10000)] if len(sys.argv) > 1 else [(16000, 256, 10000), (12950, 256, 10000), (3904, 256, 10000), (2650, 256, 10000), (3904, 256, 2048), (16000, 256, 2048), (16000, 512, 4096), (4000, 512, 10000)]
for M, N, Kd in shapes:
    A = torch.randn(M, Kd, device=dev).bfloat16(); W = (torch.randn(Kd, N, device=dev) * Kd ** -0.5).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    base = max(1, min(8, 512 // tiles, Kd // 256))
    row = []
    for split in sorted({1, 2, 3, base, min(8, base * 2), 4, 8} if len(sys.argv) > 1 else {base, min(8, base * 2), 4, 8}):
        r = []
        for mode in (0, 2, 3):
            K.gemm_configure(mode)
            kw = dict(split_k=split, c_atomic=2) if split > 1 else {}
            r.append(t(lambda: K.gemm(A, W, out, M=M, N=N, K=Kd, lda=Kd, ldb=N, ldc=N, b_kmajor=True, **kw)))
        row.append("split %d: %6.1f | %6.1f | %6.1f" % (split, *r))
    K.gemm_configure(1)
    print("%6d x %4d x %5d (python picks %d)  [128x128 | 256-row | 128-row]  " % (M, N, Kd, base) + "   ".join(row), flush=True)
