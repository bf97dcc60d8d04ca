"""The shipped row-block projection (s2t_rowblock_gemm, csrc/rowblock.hip) at the shapes tools/ubench/rowpanel_proj runs: 16 000 rows,
K = 256, N = 256 / 512 / 768 / 1024, bias only, with and without the LayerNorm prologue; buffers cycled through 6 sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
D, NB = 256, 6
g = torch.Generator().manual_seed(0)
xs = [torch.randn(M, D, generator=g).bfloat16().to(DEV) for _ in range(NB)]
gam = torch.ones(D, device=DEV); bet = torch.zeros(D, device=DEV)
xl = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)


def timeit(fn, rounds=8):
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


for N in (256, 512, 768, 1024):
    ws = [(torch.randn(N, D, generator=g) * D ** -0.5).bfloat16().to(DEV) for _ in range(NB)]
    ys = [torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(NB)]
    bias = torch.zeros(N, device=DEV)
    t_plain = timeit(lambda i: K.rowblock_gemm(xs[i], ws[i], ys[i], N=N, ldc=N, bias=bias))
    t_ln = timeit(lambda i: K.rowblock_gemm(xs[i], ws[i], ys[i], N=N, ldc=N, bias=bias, ln=(gam, bet), x_ln=xl, ln_stats=(mean, rstd)))
    print("shipped rowblock_gemm  M %5d N %4d: %6.2f us per launch bias only, %6.2f us with the LayerNorm prologue (+ x_ln, statistics saved)"
          % (M, N, t_plain, t_ln), flush=True)
