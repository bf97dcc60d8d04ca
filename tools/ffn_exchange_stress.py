"""Stress of the fused feed-forward kernels' partial-row exchange (csrc/ffn_pc.hip): the same launch repeated many times — alone, with
other buffers' launches in between (the slabs are re-used by every launch), and with compute units held by another kernel (uneven
arrival of the partners) — every word of every output compared with the first launch's.  A stale or early read of a partner's slab
shows as a block of wrong rows (MI355X_MICROARCH.md: "Test every hand-off under UNEVEN load, checking every word").
usage: python tools/ffn_exchange_stress.py [repeats=300]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

DEV = "cuda"
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 300
D, F = 256, 2048
g = torch.Generator().manual_seed(0)
bad_total = 0
for split, M in ((2, 16000), (2, 12950), (4, 6000), (8, 3904), (8, 744)):
    xs = [torch.randn(M, D, generator=g).bfloat16().to(DEV) for _ in range(3)]
    w1 = (torch.randn(F, D, generator=g) * D ** -0.5).bfloat16().to(DEV)
    w2 = (torch.randn(D, F, generator=g) * F ** -0.5).bfloat16().to(DEV)
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    b1 = torch.zeros(F, device=DEV); b2 = torch.zeros(D, device=DEV)
    gam = torch.ones(D, device=DEV); bet = torch.zeros(D, device=DEV)
    seed = torch.tensor([1], dtype=torch.int64, device=DEV)
    ys = [torch.empty(M, D, dtype=torch.bfloat16, device=DEV) for _ in range(3)]
    z = torch.empty(K.ffn_z_rows(M), F, dtype=torch.bfloat16, device=DEV)
    h = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    xl = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
    dy = torch.randn(M, D, generator=g).bfloat16().to(DEV)
    dxn = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    dz = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    K.ffn_configure(split=split)

    def fwd(i, train):
        K.ffn_fused_fwd(xs[i], w1, b1, w2, b2, ys[i], act="swish", alpha=0.5, residual=xs[i], ln=(gam, bet), x_ln=xl if train else None,
                        ln_stats=(mean, rstd) if train else None, z=z if train else None, h=h if train else None,
                        drop_h=(0.1, seed, 1) if train else None, drop_o=(0.1, seed, 2) if train else None, z_tiled_ok=train)

    def bwd():
        K.ffn_fused_bwd(dy, w2t, w1t, z, dz, dxn, act="swish", alpha=0.5, drop_h=(0.1, seed, 1), z_tiled=True)

    for mode in ("eval", "train", "bwd"):
        for load in ("alone", "interleaved", "cus_held"):
            side = torch.cuda.Stream()
            stop = torch.zeros(1, dtype=torch.int32, device=DEV)
            if mode == "bwd":
                fwd(0, True)
            ref = None
            bad = 0
            if load == "cus_held":
                with torch.cuda.stream(side):
                    K.occupy_cus(48, 400, stop=stop, stream=side)
            for r in range(REP):
                if mode == "bwd":
                    bwd()
                    out = dxn
                else:
                    fwd(0, mode == "train")
                    out = ys[0]
                cur = out.clone()
                if ref is None:
                    ref = cur
                elif not torch.equal(cur, ref):
                    bad += 1
                    if bad <= 2:
                        rows = (cur != ref).any(1).nonzero().flatten()
                        print("    MISMATCH split %d M %d %s %s rep %d: %d rows differ, first %d last %d" % (
                            split, M, mode, load, r, rows.numel(), int(rows[0]), int(rows[-1])), flush=True)
                if load == "interleaved" and mode != "bwd":
                    fwd(1 + (r & 1), mode == "train")
            stop.fill_(1)
            torch.cuda.synchronize()
            K.ffn_exchange_check()
            bad_total += bad
            print("split %d rows %5d %-5s %-11s: %d of %d repeats differ from the first launch" % (split, M, mode, load, bad, REP - 1), flush=True)
K.ffn_configure(split=0)
print("TOTAL mismatching launches:", bad_total)
sys.exit(1 if bad_total else 0)
