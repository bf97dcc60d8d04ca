#!/usr/bin/env python3
"""Epilogue cost probe: the shipped library against the -DS2T_DBG_EPI=1 (no C store) / =2 (no epilogue) builds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, torch
sys.path.insert(0, %r)
from s2t_amd import kernels as K
def bench(M, N, Kd, bias=False, act=None):
    A = torch.randn(M, Kd).to(torch.bfloat16).cuda(); B = torch.randn(N, Kd).to(torch.bfloat16).cuda()
    C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    b = torch.randn(N, device="cuda") if bias else None
    kw = dict(M=M, N=N, K=Kd, lda=Kd, ldb=Kd, ldc=N, bias=b, act=act)
    for _ in range(3): K.gemm(A, B, C, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): K.gemm(A, B, C, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3
out = []
for s in [(16000, 2048, 256), (16000, 2048, 64), (16000, 256, 256), (16000, 256, 2048), (16000, 768, 256)]:
    out.append("%%s %%6.1f" %% (s, bench(*s)))
out.append("bias+swish (16000,2048,256) %%6.1f" %% bench(16000, 2048, 256, True, "swish"))
print(" | ".join(out))
''' % ROOT
variants = [("shipped", None), ("no store", "dbg1"), ("no epilogue", "dbg2")]
if len(sys.argv) > 1:
    variants = [("shipped", None)] + [(v, v) for v in sys.argv[1:]]
for tag, lib in variants:
    env = dict(os.environ)
    if lib:
        env["S2T_HIP_LIB"] = os.path.join(ROOT, "s2t_amd", "lib", lib, "libs2t_hip.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("%-12s %s" % (tag, (r.stdout.strip() or r.stderr[-400:])), flush=True)
