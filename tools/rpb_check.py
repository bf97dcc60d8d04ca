#!/usr/bin/env python3
"""Errors of every output of s2t_relpos_attn_bwd against float64 autograd (one small problem), for experiment builds of
csrc/relpos_bwd.hip: RPB_REF=ac / bd compares dq with one branch only (libraries built with -DS2T_RPB_DBG=1 / 4)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
DEV = "cuda"
T, B, H, dk = int(os.environ.get("RPB_T", "250")), 2, 2, 64
d = H * dk; Z = B * H; bf = torch.bfloat16
g = torch.Generator().manual_seed(3)
klen = torch.tensor([T, max(1, T - 7)], dtype=torch.int32)
q = (torch.randn(B, T, d, generator=g) * 0.7).to(bf); k = (torch.randn(B, T, d, generator=g) * 0.7).to(bf)
v = (torch.randn(B, T, d, generator=g) * 0.7).to(bf); dO = (torch.randn(B, T, d, generator=g) * 0.5).to(bf)
pos = (torch.randn(2 * T - 1, d, generator=g) * 0.7).to(bf)
u = torch.randn(H, dk, generator=g) * 0.3; vb = torch.randn(H, dk, generator=g) * 0.3
scale = 1.0 / math.sqrt(dk)
pp, ud, vbd, kl = pos.to(DEV), u.reshape(-1).to(DEV), vb.reshape(-1).to(DEV), klen.to(DEV)
o = torch.empty(B, T, d, dtype=bf, device=DEV); lse = torch.empty(Z, T, dtype=torch.float32, device=DEV)
qd, kd, vd, dOd = q.to(DEV), k.to(DEV), v.to(DEV), dO.to(DEV)
K.attn_fused_fwd(qd, T * d, d, kd, T * d, d, vd, T * d, d, o, T * d, d, lse, B, H, T, T, dk, kl, False, scale, pp, d, ud, vbd, None)
du = torch.zeros(2, d, dtype=torch.float32, device=DEV)
dq = torch.zeros(B, T, d, dtype=bf, device=DEV); dkk = torch.zeros_like(dq); dv = torch.zeros_like(dq)
part = K.relpos_attn_bwd(qd, T * d, d, kd, T * d, d, vd, T * d, d, o, dOd, T * d, d, lse, dq, dkk, dv, pp, d, ud, vbd, du.view(-1),
                         du.view(-1)[d:], B, H, T, dk, kl, scale, None)
dp = torch.zeros(2 * T - 1, d, dtype=torch.float32, device=DEV)
K.relpos_dp_reduce([part], [dp], B, H, T, dk)
torch.cuda.synchronize()
qh = q.double().view(B, T, H, dk).permute(0, 2, 1, 3)
kh = k.double().view(B, T, H, dk).permute(0, 2, 1, 3).clone().requires_grad_(True)
vh = v.double().view(B, T, H, dk).permute(0, 2, 1, 3).clone().requires_grad_(True)
qu = (qh + u.double()[None, :, None, :]).to(bf).double().requires_grad_(True)
qv = (qh + vb.double()[None, :, None, :]).to(bf).double().requires_grad_(True)
ph = pos.double().view(-1, H, dk).permute(1, 2, 0).clone().requires_grad_(True)
bd_full = qv @ ph[None]
idx = (T - 1) - torch.arange(T)[:, None] + torch.arange(T)[None, :]
s = (qu @ kh.transpose(-1, -2) + torch.gather(bd_full, 3, idx[None, None].expand(B, H, T, T))) * scale
s = s.masked_fill((torch.arange(T)[None, :] >= klen.long()[:, None])[:, None, None, :], float("-inf"))
(torch.softmax(s, -1) @ vh * dO.double().view(B, T, H, dk).permute(0, 2, 1, 3)).sum().backward()
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
hd = lambda t: t.cpu().double().view(B, T, H, dk).permute(0, 2, 1, 3)
ref = {"ac": qu.grad, "bd": qv.grad}.get(os.environ.get("RPB_REF", ""), qu.grad + qv.grad)
e = (hd(dq) - ref)
print("dq %.4f  (per 32-query tile: %s)" % (rel(hd(dq), ref), " ".join("%.3f" % rel(hd(dq)[:, :, t:t + 32], ref[:, :, t:t + 32]) for t in range(0, T, 32))))
print("dk %.4f  dv %.4f  dp %.4f  du %.4f  dv %.4f" % (rel(hd(dkk), kh.grad), rel(hd(dv), vh.grad),
      rel(dp.cpu().double(), ph.grad.permute(2, 0, 1).reshape(2 * T - 1, d)), rel(du[0].cpu().double().view(H, dk), qu.grad.sum((0, 2))),
      rel(du[1].cpu().double().view(H, dk), qv.grad.sum((0, 2)))))
