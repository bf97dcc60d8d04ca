"""What the trailing LayerNorm costs in the fused FFN's row epilogue (eval flavour, 16 000 rows) against what the LayerNorm prologue
costs in the QKV row-block projection: the two sides of the "hand the LayerNorm over from the FFN to the QKV projection" trade
(VERDICT round 5, item 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

DEV = "cuda"
M, D, F, NB = 16000, 256, 2048, 12
g = torch.Generator().manual_seed(0)
xs = [torch.randn(M, D, generator=g).bfloat16().to(DEV) for _ in range(NB)]
w1 = [(torch.randn(F, D, generator=g) * D ** -0.5).bfloat16().to(DEV) for _ in range(NB)]
w2 = [(torch.randn(D, F, generator=g) * F ** -0.5).bfloat16().to(DEV) for _ in range(NB)]
wq = [(torch.randn(3 * D, D, generator=g) * D ** -0.5).bfloat16().to(DEV) for _ in range(NB)]
b1 = torch.zeros(F, device=DEV); b2 = torch.zeros(D, device=DEV); bq = torch.zeros(3 * D, device=DEV)
gam = torch.ones(D, device=DEV); bet = torch.zeros(D, device=DEV)
y = torch.empty(M, D, dtype=torch.bfloat16, device=DEV); yl = torch.empty_like(y)
qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=DEV)
em = torch.empty(M, device=DEV); er = torch.empty(M, device=DEV)


def timeit(fn, rounds=6):
    for i in range(NB):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rounds):
            for i in range(NB):
                fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (rounds * NB) * 1e3)
    return best


def ffn(i, end):
    K.ffn_fused_fwd(xs[i], w1[i], b1, w2[i], b2, y, act="swish", alpha=0.5, residual=xs[i], ln=(gam, bet),
                    end_ln=(gam, bet) if end else None, y_ln=yl if end else None, end_stats=(em, er) if end else None)


t0, t1 = timeit(lambda i: ffn(i, False)), timeit(lambda i: ffn(i, True))
q0 = timeit(lambda i: K.rowblock_gemm(xs[i], wq[i], qkv, N=3 * D, ldc=3 * D, bias=bq))
q1 = timeit(lambda i: K.rowblock_gemm(xs[i], wq[i], qkv, N=3 * D, ldc=3 * D, bias=bq, ln=(gam, bet)))
print("fused FFN eval: %.2f us without, %.2f us with a trailing LayerNorm (+ y_ln, statistics): +%.2f us" % (t0, t1, t1 - t0))
print("QKV projection: %.2f us without, %.2f us with the LayerNorm prologue: +%.2f us" % (q0, q1, q1 - q0))
print("hand-over FFN -> QKV would move %.2f us out of the projection and %.2f us into the FFN: net %.2f us per layer" % (q1 - q0, t1 - t0, (q1 - q0) - (t1 - t0)))
