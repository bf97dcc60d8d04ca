#!/usr/bin/env python3
"""s_memtime stamps of the relative-position dQ kernel (library built with -DS2T_ATT_DBG=64): cycles between the stamped
points for the four waves of two workgroups.  Points: start | prologue done | per key block: barrier 1, tiles stored +
barrier 2, scores (QK^T + band), [dP + dS], dBD stores, dQ MFMAs | end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
dev = "cuda"
B, H, T, dk = 64, 4, 250, 64
d = H * dk
qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
dO = (torch.randn(B * T, d, device=dev) * 0.1).to(torch.bfloat16)
lse = torch.empty(B * H * T, device=dev)
delta = torch.zeros(B * H * T + 8 * 40 * 2 + 16, device=dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
n_pos = 2 * T - 1
p = (torch.randn(n_pos, d, device=dev) * 0.5).to(torch.bfloat16)
pu, pv = torch.randn(d, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1
seed = torch.tensor([5], dtype=torch.int64, device=dev)
dqkv = torch.empty(B * T, 3 * d, dtype=torch.bfloat16, device=dev)
ldB = (n_pos + 7) // 8 * 8
dbd = torch.zeros(H, B, T, ldB, dtype=torch.bfloat16, device=dev)
kw = dict(pos_p=p, p_sr=d, pos_u=pu, pos_v=pv, drop=(0.1, seed, 3))
K.attn_fused_fwd(q, T * 3 * d, 3 * d, k, T * 3 * d, 3 * d, v, T * 3 * d, 3 * d, o, T * d, d, lse, B, H, T, T, dk, lens, False, dk ** -0.5, **kw)
for _ in range(3):
    K.attn_fused_bwd(q, T * 3 * d, 3 * d, k, T * 3 * d, 3 * d, v, T * 3 * d, 3 * d, o, dO, T * d, d, lse, delta, dqkv, dqkv[:, d:], dqkv[:, 2 * d:], dbd, ldB, B, H, T, T, dk, lens, False, dk ** -0.5, dbd_band_only=True, **kw)
torch.cuda.synchronize()
raw = delta[B * H * T:B * H * T + 8 * 40 * 2].view(torch.int64).cpu().view(8, 40)
names = ["prologue"] + sum([["barrier1", "store+barrier2", "scores", "dP+dS", "dBD stores", "dQ mfma"] for _ in range(4)], []) 
for wv in range(8):
    st = [int(x) for x in raw[wv] if int(x) != 0]
    if len(st) < 3:
        continue
    d_ = [st[i + 1] - st[i] for i in range(len(st) - 1)]
    print("wg %d wave %d: total %d clk (100 MHz ticks x?) :" % (wv // 4, wv % 4, st[-1] - st[0]), " ".join("%d" % x for x in d_))
