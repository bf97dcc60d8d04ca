#!/usr/bin/env python3
"""s2t_relpos_glue alone at the headline shape (61 utterances x 4 heads, T' = 250): HIP-event time per call, fresh dbd per call
(rotating buffers: the previous kernel of the real step leaves dbd partly in the caches, a hot loop would leave all of it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
B, H, Tq, dk = int(os.environ.get("PROBE_B", 61)), 4, int(os.environ.get("PROBE_T", 250)), 64
d = H * dk; n_pos = 2 * Tq - 1; ldb = (n_pos + 7) // 8 * 8
dev = "cuda"
NB = 6
dbd = [(torch.randn(H, B, Tq, ldb, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NB)]
p = (torch.randn(n_pos, d, device=dev) * 0.7).to(torch.bfloat16)
qv = (torch.randn(B * Tq, d, device=dev) * 0.6).to(torch.bfloat16)
dq = [torch.zeros(B * Tq, 3 * d, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
ws = torch.zeros(32, 2, d, device=dev); dp = torch.zeros(n_pos, d, device=dev)
def f(i):
    K.relpos_glue(dbd[i % NB], ldb, p, d, qv, dq[i % NB], Tq * 3 * d, 3 * d, ws.view(-1), ws.view(-1)[d:], dp, B, H, Tq, dk, replicas=32, replica_stride=2 * d)
for i in range(6): f(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 60
e0.record()
for i in range(N): f(i)
e1.record(); torch.cuda.synchronize()
print("%s: glue + reduce %.1f us per call (dbd %.1f MB)" % (os.environ.get("S2T_HIP_LIB", "default").split("/")[-2:][0], e0.elapsed_time(e1) / N * 1e3, dbd[0].numel() * 2 / 1e6))
