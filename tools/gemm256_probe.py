#!/usr/bin/env python3
"""s2t_gemm's 256 x 256 LDS-DMA path against the 128 x 128 path (s2t_gemm_configure 2 / 0) in one process: equality of the
results and interleaved timings on the Linear shapes of the configurations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def run(M, N, Kd, out_dtype=torch.bfloat16, bias=False, act=None, residual=False, rounds=15, check=True, bkm=False):
    A = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(torch.bfloat16)
    if bkm:
        B = B.t().contiguous()  # [K][N]
    b = torch.randn(N, device=dev) if bias else None
    R = torch.randn(M, N, device=dev).to(out_dtype) if residual else None
    outs = {}
    ts = {0: [], MODE_NEW: []}
    for mode in (0, MODE_NEW):
        K.gemm_configure(mode)
        C = torch.empty(M, N // 2 if act == "glu" else N, device=dev, dtype=out_dtype)
        kw = dict(M=M, N=N, K=Kd, lda=Kd, ldb=N if bkm else Kd, ldc=N // 2 if act == "glu" else N, bias=b, act=act, residual=R, ldr=N if residual else 0, b_kmajor=bkm)
        K.gemm(A, B, C, **kw)
        outs[mode] = C
    torch.cuda.synchronize()
    same = torch.equal(outs[0], outs[MODE_NEW])
    err = None
    if check and M * N <= 64000 * 2048:
        ref = A.float() @ (B.float() if bkm else B.float().t())
        if bias: ref += b
        if act == "relu": ref = ref.relu()
        if residual: ref += R.float()
        err = float((outs[MODE_NEW].float() - ref).abs().max() / ref.abs().max())
    for r in range(rounds):
        for mode in (0, MODE_NEW):
            K.gemm_configure(mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                K.gemm(A, B, outs[mode], **kw)
            e1.record()
            torch.cuda.synchronize()
            ts[mode].append(e0.elapsed_time(e1) * 250)
    t0, t2 = sorted(ts[0])[rounds // 2], sorted(ts[MODE_NEW])[rounds // 2]
    fl = 2.0 * M * N * Kd
    print("M%6d N%6d K%5d %s%s%s%s%s : old %7.1f us %6.0f TF/s | 256 %7.1f us %6.0f TF/s  x%.2f  equal=%s err=%s" % (
        M, N, Kd, "f32" if out_dtype == torch.float32 else "bf16", " bias" if bias else "", " " + act if act else "",
        " res" if residual else "", " Bkm" if bkm else "", t0, fl / t0 / 1e6, t2, fl / t2 / 1e6, t0 / t2, same, "%.1e" % err if err is not None else "-"), flush=True)
    K.gemm_configure(1)
    return same


shapes = [
    (1024, 512, 512, {}), (300, 1000, 256, dict(bias=True)), (777, 520, 128, dict(bias=True, act="relu", residual=True)),
    (64000, 2048, 512, dict(bias=True, act="relu")), (64000, 512, 2048, dict(bias=True, residual=True)),
    (64000, 1536, 512, dict(bias=True)), (64000, 512, 512, dict(bias=True)), (64000, 1024, 512, {}),
    (64000, 10000, 512, dict(bias=True, out_dtype=torch.float32, check=False)), (64000, 10000, 512, dict(bias=True, check=False)),
    (16000, 10000, 256, dict(bias=True)), (16000, 3072, 256, dict(bias=True)), (16000, 768, 256, dict(bias=True)),
    (16000, 2048, 256, dict(bias=True, act="relu")), (16000, 256, 2048, dict(bias=True, residual=True)),
    (13100, 10000, 256, dict(bias=True)), (3904, 10000, 256, {}), (8192, 8192, 8192, dict(rounds=5, check=False)),
    (4096, 4096, 4096, dict(rounds=7)),
    (1000, 520, 200, dict(bias=True)), (64000, 512, 10000, dict(bkm=True, residual=True, check=False)), (64000, 2048, 512, dict(bkm=True)),
    (64000, 512, 2048, dict(bkm=True)), (16000, 10000, 256, dict(bkm=True)), (8192, 8192, 8192, dict(bkm=True, rounds=5, check=False)),
    (1000, 520, 200, dict(bkm=True, bias=True)), (64000, 512, 10000, dict(residual=True, check=False)),
    (64000, 1024, 512, dict(act="glu", bias=True, check=False)), (32000, 1024, 400, dict(act="glu", bias=True, check=False)),
    (128000, 4096, 400, dict(act="glu", bias=True, check=False)),
]
import sys
MODE_NEW = 3 if (len(sys.argv) > 1 and sys.argv[1] == "rows128") else 2   # rows128: the 128-row large tiles against the 128 x 128 path
if MODE_NEW == 3:
    shapes = [(16000, 512, 512, dict(bias=True)), (16000, 512, 2048, dict(bias=True, residual=True)), (16000, 2048, 512, dict(bias=True, act="relu")),
              (16000, 1536, 512, {}), (16000, 512, 256, {}), (16000, 1024, 256, {}), (13100, 512, 512, {}), (16000, 512, 512, dict(bkm=True)),
              (16000, 512, 2048, dict(bkm=True)), (16000, 512, 10000, dict(bkm=True)), (8000, 1024, 512, {}), (32000, 256, 512, {}),
              (2650, 2048, 256, {}), (16000, 768, 256, {}), (64000, 2048, 512, dict(bias=True))]
if len(sys.argv) > 1 and sys.argv[1] == "border":  # shapes around the automatic mode's tile-count threshold
    shapes = [(16000, 512, 512, {}), (16000, 512, 2048, {}), (8000, 1024, 256, {}), (4000, 2048, 512, {}), (16000, 768, 256, {}),
              (16000, 1024, 256, {}), (13100, 512, 256, {}), (13100, 768, 256, {}), (2650, 10000, 256, {}), (2650, 2048, 256, {}),
              (8192, 512, 512, {}), (32000, 256, 256, {}), (32000, 512, 256, {}), (64000, 256, 512, {}), (13100, 3072, 256, {})]
ok = True
for M, N, Kd, kw in shapes:
    ok &= run(M, N, Kd, **kw)
print("ALL EQUAL" if ok else "MISMATCH")
