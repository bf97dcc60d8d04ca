#!/usr/bin/env python3
"""Fused attention kernels at the encoder shape (B=64, H=4, T'=250, dk=64): relative-position vs plain, dropout on/off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
dev = "cuda"
B, H, T, dk = 64, 4, 250, 64
d = H * dk
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
dO = (torch.randn(B * T, d, device=dev) * 0.1).to(torch.bfloat16)
lse = torch.empty(B * H * T, device=dev); delta = torch.empty(B * H * T, device=dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
n_pos = 2 * T - 1
p = (torch.randn(n_pos, d, device=dev) * 0.5).to(torch.bfloat16)
pu, pv = torch.randn(d, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1
seed = torch.tensor([5], dtype=torch.int64, device=dev)
dqkv = torch.empty(B * T, 3 * d, dtype=torch.bfloat16, device=dev)
ldB = (n_pos + 7) // 8 * 8
dbd = torch.zeros(H, B, T, ldB, dtype=torch.bfloat16, device=dev)
ld_t = (16 + n_pos + 96 + 7) // 8 * 8
ptb = torch.zeros(d, ld_t, dtype=torch.bfloat16, device=dev); ptb[:, 16:16 + n_pos] = p.t()
du, dvv = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
FUSE = os.environ.get("PROBE_FUSE_V", "1") == "1"
for rel in (False, True):
    for drop in (None, (0.1, seed, 3)):
        kw = dict(pos_p=p if rel else None, p_sr=d, pos_u=pu if rel else None, pos_v=pv if rel else None, drop=drop)
        f = lambda: K.attn_fused_fwd(q, T * 3 * d, 3 * d, k, T * 3 * d, 3 * d, v, T * 3 * d, 3 * d, o, T * d, d, lse, B, H, T, T, dk, lens, False, dk ** -0.5, **kw)
        b = lambda: K.attn_fused_bwd(q, T * 3 * d, 3 * d, k, T * 3 * d, 3 * d, v, T * 3 * d, 3 * d, o, dO, T * d, d, lse, delta, dqkv, dqkv[:, d:], dqkv[:, 2 * d:], dbd if rel else None, ldB, B, H, T, T, dk, lens, False, dk ** -0.5, dbd_band_only=rel, **(dict(kw, pos_pt=ptb[:, 16:], pt_ld=ld_t, dpos_u=du, dpos_v=dvv) if (rel and FUSE) else kw))
        print("rel=%-5s dropout=%-5s fwd %6.1f us   bwd (delta + dq + dkv) %6.1f us" % (rel, drop is not None, t(f), t(b)), flush=True)
