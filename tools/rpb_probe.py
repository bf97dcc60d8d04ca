#!/usr/bin/env python3
"""s2t_relpos_attn_bwd at the bench's shape (64 utterances x 4 heads, T' = 250, packed rows at the bench's fill): time per launch;
with a library built with -DS2T_RPB_DBG=64 (tools/dbg_variant.sh, S2T_HIP_LIB) the per-phase clock stamps of workgroup 0."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K, rows as Rows
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, H, T, dk = 64, 4, 250, 64
d = H * dk
bf = torch.bfloat16
g = torch.Generator().manual_seed(1)
lens = sorted([T] + [int(torch.randint(int(0.6 * T), T + 1, (1,), generator=g)) for _ in range(B - 1)], reverse=True)
if os.environ.get("RPB_FULL") == "1":
    lens = [T] * B
kl = torch.tensor(lens, dtype=torch.int32, device=dev)
rows = Rows.attach(kl, B, T, 0)
n = sum(lens)
mk = lambda r, c, s=0.7: (torch.randn(r, c, device=dev) * s).to(bf)
qkv = mk(n + 8, 3 * d)
q, k, v = qkv, qkv[:, d:], qkv[:, 2 * d:]
dqkv = torch.empty_like(qkv)
o, dO = mk(n + 8, d), mk(n + 8, d, 0.5)
pos = mk(2 * T - 1, d)
u, vb = torch.randn(d, device=dev) * 0.3, torch.randn(d, device=dev) * 0.3
lse = torch.randn(B * H, T, device=dev) + 5.0
seed = torch.full((1,), 7, dtype=torch.int64, device=dev)
drop = (0.1, seed, 3) if os.environ.get("RPB_NODROP") != "1" else None
ws = torch.zeros(32, 2, d, device=dev)
K._scratch("relpos_dp_part", (B * (2 * T - 1) * d + 1) // 2 + 4096, dev)
run = lambda: K.relpos_attn_bwd(q, 0, 3 * d, k, 0, 3 * d, v, 0, 3 * d, o, dO, 0, d, lse, dqkv, dqkv[:, d:], dqkv[:, 2 * d:], pos, d, u, vb,
                                ws.view(-1), ws.view(-1)[d:], B, H, T, dk, rows, 1 / math.sqrt(dk), drop, replicas=32,
                                replica_stride=2 * d, rows=rows)
for _ in range(3):
    part = run()
torch.cuda.synchronize()
ts = []
for r in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 100)
print("relpos_attn_bwd %d rows (fill %.2f): %.1f us" % (n, n / (B * T), sorted(ts)[3]), flush=True)
st = part.view(torch.int64)[(B * (2 * T - 1) * d * 2) // 8:][:8 * 16].cpu().view(8, 16)
if int(st[0, 0]) > 0 and 0 < int(st[0, 1]) - int(st[0, 0]) < 10 ** 9:
    # stamps: 0 loop start | second tile: 1 after the top barrier, 2 after phase A + barrier, 3 end of phase B, 4 after the barrier,
    # 5 after product (3), 6 after product (1), 7 after the dq store + product (2) | 8 after the loop
    names = ["tile0", "A+bar", "B", "wait", "(3)", "(1)", "st+(2)", "tiles2.."]
    for w in range(8):
        s = st[w].tolist()
        print("wave %d:" % w, "  ".join("%s %d" % (names[i], s[i + 1] - s[i]) for i in range(len(names)) if s[i + 1] > 0))
