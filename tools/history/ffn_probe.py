"""Fused FFN row-block kernel vs the unfused LayerNorm + two GEMMs, encoder shape of the headline configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

DEV = "cuda"
M, d, F = int(os.environ.get("M", 16000)), 256, int(os.environ.get("F", 2048))
g = torch.Generator().manual_seed(0)
x = torch.randn(M, d, generator=g).bfloat16().to(DEV)
w1 = (torch.randn(F, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
w2 = (torch.randn(d, F, generator=g) * F ** -0.5).bfloat16().to(DEV)
b1 = (0.1 * torch.randn(F, generator=g)).to(DEV)
b2 = (0.1 * torch.randn(d, generator=g)).to(DEV)
gam = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
bet = (0.1 * torch.randn(d, generator=g)).to(DEV)
seed = torch.tensor([1], dtype=torch.int64, device=DEV)
xl = torch.empty_like(x); mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
h = torch.empty(M, F, dtype=torch.bfloat16, device=DEV); z = torch.empty_like(h)
y = torch.empty_like(x); yl = torch.empty_like(x)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def unfused(train, act):
    dh = (0.1, seed, 1) if train else None
    do = (0.1, seed, 2) if train else None
    K.layernorm_fwd(x, gam, bet, xl, mean, rstd, M, d)
    K.gemm(xl, w1, h, M=M, N=F, K=d, lda=d, ldb=d, ldc=F, bias=b1, act=act, preact=z if train else None, ldp=F, drop=dh)
    K.gemm(h, w2, y, M=M, N=d, K=F, lda=F, ldb=F, ldc=d, bias=b2, alpha=0.5, residual=x, ldr=d, drop=do)


def fused(train, act, eln=False, drop=True):
    dh = (0.1, seed, 1) if (train and drop) else None
    do = (0.1, seed, 2) if (train and drop) else None
    K.ffn_fused_fwd(x, w1, b1, w2, b2, y, act=act, alpha=0.5, residual=x, ln=(gam, bet),
                    x_ln=xl if train else None, ln_stats=(mean, rstd) if train else None, z=z if train else None,
                    h=h if train else None, drop_h=dh, drop_o=do, end_ln=(gam, bet) if eln else None,
                    y_ln=yl if eln else None)


flop = 4.0 * M * F * d
for act in ("swish", "relu"):
    for train in (False, True):
        tu = timeit(lambda: unfused(train, act))
        tf = timeit(lambda: fused(train, act))
        print("act=%s train=%d  unfused %.1f us (%.0f TF/s)   fused %.1f us (%.0f TF/s)" % (
            act, train, tu, flop / tu / 1e6, tf, flop / tf / 1e6), flush=True)
tf = timeit(lambda: fused(True, "swish", False, False))
print("fused train swish without dropout: %.1f us" % tf)
tf = timeit(lambda: fused(False, "swish", True))
print("fused eval swish + end LayerNorm: %.1f us" % tf)
