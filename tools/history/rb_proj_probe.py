"""Row-block projection kernel vs LayerNorm + s2t_gemm for the K = 256 projections of the headline encoder layer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

DEV = "cuda"
M, d = 16000, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(M, d, generator=g).bfloat16().to(DEV)
gam = torch.ones(d, device=DEV); bet = torch.zeros(d, device=DEV)
xl = torch.empty_like(x); mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
lens = torch.full((64,), 250, dtype=torch.int32, device=DEV)


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


for name, N, act, ln, res in (("qkv", 768, None, True, False), ("pw1+glu", 512, "glu", True, False),
                              ("out-proj", 256, None, False, True), ("pw2", 256, None, False, True)):
    w = (torch.randn(N, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
    bias = torch.zeros(N, device=DEV) if act != "glu" else None
    nout = N // 2 if act == "glu" else N
    out = torch.empty(M, nout, dtype=torch.bfloat16, device=DEV)
    resid = torch.randn(M, nout, generator=g).bfloat16().to(DEV) if res else None

    def unfused():
        src = x
        if ln:
            K.layernorm_fwd(x, gam, bet, xl, mean, rstd, M, d, 1e-5, lens if act == "glu" else None, 250)
            src = xl
        K.gemm(src, w, out, M=M, N=N, K=d, lda=d, ldb=d, ldc=nout, bias=bias, act=act, residual=resid, ldr=nout)

    def fused():
        K.rowblock_gemm(x, w, out, N=N, ldc=nout, bias=bias, act=act, residual=resid, ldr=nout,
                        ln=(gam, bet) if ln else None, ln_lens=lens if (ln and act == "glu") else None, ln_T=250)

    tu, tf = timeit(unfused), timeit(fused)
    fl = 2.0 * M * N * d
    print("%-9s N=%4d  unfused %.1f us (%.0f TF/s)   rowblock %.1f us (%.0f TF/s)" % (name, N, tu, fl / tu / 1e6, tf, fl / tf / 1e6), flush=True)
