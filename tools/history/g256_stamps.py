#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the large-tile GEMM (a -DS2T_G256_DBG=16 build: tools/g256_dbg.sh; S2T_HIP_LIB names it):
cycles of wave 0 per segment, median over workgroups."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K, _lib as L
dev = torch.device("cuda", 0)
torch.manual_seed(0)
K.gemm_configure(2)
for M, N, Kd in [(64000, 2048, 512), (64000, 512, 2048), (16000, 10000, 256), (8192, 8192, 8192)]:
    A = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(torch.bfloat16)
    Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.zeros(256 * 8, device=dev)
    a = L.GemmArgs()
    a.dtype = a.c_dtype = L.dtype_id(torch.bfloat16)
    a.M, a.N, a.K = M, N, Kd
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = A.data_ptr(), Kd, B.data_ptr(), Kd, Cm.data_ptr(), N
    a.batch = a.zdiv = a.split_k = 1
    a.alpha = 1.0
    a.colsum_a = ws.data_ptr()
    for _ in range(3):
        L.check(L.lib().s2t_gemm(C.byref(a), L.stream_ptr()), "s2t_gemm")
    torch.cuda.synchronize()
    w = ws.view(256, 8).cpu()
    med = w.median(0).values
    steps = med[5].item()
    tiles = steps / (Kd // 64)
    print("%dx%dx%d: steps %.0f tiles %.1f | wait fresh %.0f  second %.0f  other %.0f (per step %.0f) | multiply %.0f per step | epilogue %.0f per tile | total %.0f kcyc" % (
        M, N, Kd, steps, tiles, med[0] / max(tiles - 1, 1), med[1] / max(tiles - 1, 1), med[2], med[2] / max(steps - 2 * (tiles - 1), 1), med[3] / steps, med[4] / tiles,
        (med[0] + med[1] + med[2] + med[3] + med[4]) / 1e3), flush=True)
