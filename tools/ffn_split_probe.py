"""The fused feed-forward flavours at the DECODER's row count (64 x 61 = 3 904 target rows, relu, dropout 0.1) for every forced split
(workgroups per 128-row block: 1, 2, 4, 8; 0 = the library's own choice), buffers cycled through 12 sets: which split should the
decoder run (the eight-part forms move 3.2 x their algorithmic bytes at 6 % of the MFMA peak)?
usage: python tools/ffn_split_probe.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3904
D, F, NB = 256, 2048, 12
g = torch.Generator().manual_seed(0)
xs = [torch.randn(M, D, generator=g).bfloat16().to(DEV) for _ in range(NB)]
w1 = [(torch.randn(F, D, generator=g) * D ** -0.5).bfloat16().to(DEV) for _ in range(NB)]
w2 = [(torch.randn(D, F, generator=g) * F ** -0.5).bfloat16().to(DEV) for _ in range(NB)]
w1t = [w.t().contiguous() for w in w1]
w2t = [w.t().contiguous() for w in w2]
b1 = torch.zeros(F, device=DEV); b2 = torch.zeros(D, device=DEV)
gam = torch.ones(D, device=DEV); bet = torch.zeros(D, device=DEV)
seed = torch.tensor([1], dtype=torch.int64, device=DEV)
zs = [torch.empty(K.ffn_z_rows(M), F, dtype=torch.bfloat16, device=DEV) for _ in range(NB)]
hs = [torch.empty(M, F, dtype=torch.bfloat16, device=DEV) for _ in range(NB)]
xl = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
y = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
dy = torch.randn(M, D, generator=g).bfloat16().to(DEV)
dxn = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)


def fwd(i, train):
    K.ffn_fused_fwd(xs[i], w1[i], b1, w2[i], b2, y, act="relu", alpha=1.0, residual=xs[i], ln=(gam, bet), x_ln=xl if train else None,
                    ln_stats=(mean, rstd) if train else None, z=zs[i] if train else None, h=hs[i] if train else None,
                    drop_h=(0.1, seed, 1) if train else None, drop_o=(0.1, seed, 2) if train else None, z_tiled_ok=train)


def bwd(i):
    K.ffn_fused_bwd(dy, w2t[i], w1t[i], zs[i], hs[i], dxn, act="relu", alpha=1.0, drop_h=(0.1, seed, 1), z_tiled=True)


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


_, old, _ = K.ffn_configure()
print("rows %d  (split 0 = the library's choice)" % M)
for split in (0, 1, 2, 4, 8):
    K.ffn_configure(split=split)
    try:
        te, tf, tb = timeit(lambda i: fwd(i, False)), timeit(lambda i: fwd(i, True)), timeit(bwd)
        K.ffn_exchange_check()
        print("split %d: eval %6.1f us   training forward %6.1f us   backward %6.1f us" % (split, te, tf, tb), flush=True)
    except Exception as e:  # noqa: BLE001
        print("split %d: %s" % (split, e), flush=True)
K.ffn_configure(split=old)
