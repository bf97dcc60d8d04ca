#!/usr/bin/env python3
"""Plain attention backward, the two routes side by side at the shapes of the recipes (64 utterances x 4 heads, dropout 0.1):
s2t_attn_fused_bwd (two kernels) against s2t_attn_bwd_one_pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, H, dk = 64, 4, 64
d = H * dk
bf = torch.bfloat16
seed = torch.full((1,), 7, dtype=torch.int64, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000 / n)
    return sorted(ts)[2]
for Tq, Tk, causal in [(250, 250, False), (61, 61, True), (61, 250, False), (128, 128, False)]:
    mk = lambda T_, s=0.7: (torch.randn(B, T_, d, device=dev) * s).to(bf)
    q, k, v, o, dO = mk(Tq), mk(Tk), mk(Tk), mk(Tq), mk(Tq, 0.5)
    lse = torch.randn(B * H, Tq, device=dev) + 5
    kl = torch.full((B,), Tk, dtype=torch.int32, device=dev)
    kl[1::2] = int(0.7 * Tk)
    dq, dkk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B * H, Tq, device=dev)
    drop = (0.1, seed, 3)
    scale = dk ** -0.5
    a = t(lambda: K.attn_fused_bwd(q, Tq * d, d, k, Tk * d, d, v, Tk * d, d, o, dO, Tq * d, d, lse, delta, dq, dkk, dv, None, 0, B, H, Tq,
                                   Tk, dk, kl, causal, scale, None, 0, None, None, drop))
    b = t(lambda: K.attn_bwd_one_pass(q, Tq * d, d, k, Tk * d, d, v, Tk * d, d, o, dO, Tq * d, d, lse, dq, dkk, dv, B, H, Tq, Tk, dk, kl,
                                      causal, scale, drop))
    print("Tq %3d Tk %3d causal %d: two kernels %.1f us, one pass %.1f us" % (Tq, Tk, causal, a, b), flush=True)
