"""Fused vs unfused FFN (training flavour) when every call writes its saves to DIFFERENT memory (as inside a model: 24 blocks
x 131 MB per step), instead of re-using one buffer that stays resident in the 256 MiB Infinity Cache."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

DEV = "cuda"
M, d, F = int(os.environ.get("PROBE_M", "16000")), 256, int(os.environ.get("PROBE_F", "2048"))
NB = 12 if M <= 16000 else 3
g = torch.Generator().manual_seed(0)
xs = [torch.randn(M, d, generator=g).bfloat16().to(DEV) for _ in range(NB)]
ws1 = [(torch.randn(F, d, generator=g) * d ** -0.5).bfloat16().to(DEV) for _ in range(NB)]
ws2 = [(torch.randn(d, F, generator=g) * F ** -0.5).bfloat16().to(DEV) for _ in range(NB)]
b1 = torch.zeros(F, device=DEV); b2 = torch.zeros(d, device=DEV)
gam = torch.ones(d, device=DEV); bet = torch.zeros(d, device=DEV)
seed = torch.tensor([1], dtype=torch.int64, device=DEV)
zs = [torch.empty(K.ffn_z_rows(M), F, dtype=torch.bfloat16, device=DEV) for _ in range(NB)]
hs = [torch.empty(M, F, dtype=torch.bfloat16, device=DEV) for _ in range(NB)]
xl = torch.empty_like(xs[0]); mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
y = torch.empty_like(xs[0])


def unfused(i, train):
    dh = (0.1, seed, 1) if train else None
    do = (0.1, seed, 2) if train else None
    K.layernorm_fwd(xs[i], gam, bet, xl, mean, rstd, M, d)
    K.gemm(xl, ws1[i], hs[i], M=M, N=F, K=d, lda=d, ldb=d, ldc=F, bias=b1, act="swish", preact=zs[i] if train else None, ldp=F, drop=dh)
    K.gemm(hs[i], ws2[i], y, M=M, N=d, K=F, lda=F, ldb=F, ldc=d, bias=b2, alpha=0.5, residual=xs[i], ldr=d, drop=do)


NODROP = os.environ.get("PROBE_NODROP", "0") == "1"
TILED = os.environ.get("PROBE_TILED", "1") == "1"  # z in the tiled layout (what the model does when both passes are fused)


def fused(i, train):
    dh = (0.1, seed, 1) if (train and not NODROP) else None
    do = (0.1, seed, 2) if (train and not NODROP) else None
    K.ffn_fused_fwd(xs[i], ws1[i], b1, ws2[i], b2, y, act="swish", alpha=0.5, residual=xs[i], ln=(gam, bet),
                    x_ln=xl if train else None, ln_stats=(mean, rstd) if train else None, z=zs[i] if train else None,
                    h=hs[i] if train else None, drop_h=dh, drop_o=do, z_tiled_ok=train and TILED)


def timeit(fn, train, rounds=4):
    for i in range(NB):
        fn(i, train)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        for i in range(NB):
            fn(i, train)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (rounds * NB) * 1e3


for train in (False, True):
    print("train=%d  cycling %d buffer sets:  unfused %.1f us   fused %.1f us" % (train, NB, timeit(unfused, train), timeit(fused, train)), flush=True)

# ---- backward: s2t_ffn_fused_bwd vs the two dgrad GEMMs it replaces, dz going to different memory each call
dy = torch.randn(M, d, generator=g).bfloat16().to(DEV)
w2ts = [w.t().contiguous() for w in ws2]
w1ts = [w.t().contiguous() for w in ws1]
dxn = torch.empty_like(dy)


def bwd_unfused(i, train):
    dh = (0.1, seed, 1)
    K.gemm(dy, ws2[i], hs[i], M=M, N=F, K=d, lda=d, ldb=F, ldc=F, b_kmajor=True, alpha=0.5, dact_z=zs[i], ldz=F, dact="swish", drop=dh)
    K.gemm(hs[i], ws1[i], dxn, M=M, N=d, K=F, lda=F, ldb=d, ldc=d, b_kmajor=True)


def bwd_fused(i, train):
    K.ffn_fused_bwd(dy, w2ts[i], w1ts[i], zs[i], hs[i], dxn, act="swish", alpha=0.5, drop_h=None if NODROP else (0.1, seed, 1), z_tiled=TILED)


print("backward, cycling %d buffer sets:  two dgrad GEMMs %.1f us   fused %.1f us" % (NB, timeit(bwd_unfused, True), timeit(bwd_fused, True)), flush=True)
