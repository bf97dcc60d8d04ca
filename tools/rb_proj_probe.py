"""Row-block projection kernels of the headline encoder layer as the TRAINING step calls them (saves, dropout, masks), timed
one by one with HIP events; S2T_HIP_LIB selects the library (tools/dbg_variant.sh), so two runs on one box give an A/B.
usage (GPU box): python tools/rb_proj_probe.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K

DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
d = 256
g = torch.Generator().manual_seed(0)
x = torch.randn(M, d, generator=g).bfloat16().to(DEV)
gam = torch.ones(d, device=DEV)
bet = torch.zeros(d, device=DEV)
xl = torch.empty_like(x)
mean = torch.empty(M, device=DEV)
rstd = torch.empty(M, device=DEV)
B = M // 250
lens = torch.randint(150, 251, (B,), generator=g).to(torch.int32).to(DEV)
seed = torch.tensor([1234], dtype=torch.int64, device=DEV)


def timeit(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = (("qkv", 768, None, True, False, False), ("pw1+glu", 512, "glu", True, False, False),
         ("out-proj", 256, None, False, True, True), ("pw2", 256, None, False, True, True))
for name, N, act, ln, res, drop in cases:
    w = (torch.randn(N, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
    bias = torch.zeros(N, device=DEV)
    nout = N // 2 if act == "glu" else N
    out = torch.empty(M, nout, dtype=torch.bfloat16, device=DEV)
    resid = torch.randn(M, nout, generator=g).bfloat16().to(DEV) if res else None
    z = torch.empty(M, N, dtype=torch.bfloat16, device=DEV) if act == "glu" else None
    sc = torch.ones(d, device=DEV)
    sh = torch.zeros(d, device=DEV)

    def fused():
        if ln:
            K.rowblock_gemm(x, w, out, N=N, ldc=nout, bias=bias, act=act, preact=z, ldp=N if z is not None else 0,
                            ln=(gam, bet), ln_lens=lens if act == "glu" else None, ln_T=250, x_ln=xl, ln_stats=(mean, rstd))
        elif name == "pw2":
            K.rowblock_gemm(x, w, out, N=N, ldc=nout, bias=bias, residual=resid, ldr=nout, pre=(sc, sh, "swish"), ln_lens=lens,
                            ln_T=250, x_ln=xl, drop=(0.1, seed, 7))
        else:
            K.rowblock_gemm(x, w, out, N=N, ldc=nout, bias=bias, residual=resid, ldr=nout, drop=(0.1, seed, 5))

    tf = timeit(fused)
    fl = 2.0 * M * N * d
    print("%-9s N=%4d  rowblock %.1f us (%.0f TF/s)" % (name, N, tf, fl / tf / 1e6), flush=True)

    # in-kernel clock stamps of workgroups 0 and 100 (an -DS2T_RB_DBG=64 build: tools/dbg_variant.sh rbgdbg rowblock.hip -DS2T_RB_DBG=64)
    from s2t_amd import _lib as L
    if hasattr(L.lib(), "s2t_rbg_dbg_read"):
        import ctypes
        torch.cuda.synchronize()
        fused()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 64)()
        L.lib().s2t_rbg_dbg_read(buf)
        for wg in (0, 1):
            st = [buf[32 * wg + i] for i in range(32)]
            t0 = st[0]
            names = {1: "live", 2: "prologue issued", 3: "loads landed", 4: "sync", 5: "fragments", 28: "last read-out", 29: "stores drained"}
            line = []
            prev = t0
            for i in range(1, 30):
                if st[i] == 0 or st[i] < t0:
                    continue
                line.append("%s +%d" % (names.get(i, "c%d" % (i - 6)), st[i] - prev))
                prev = st[i]
            print("   wg%-3d total %d clk: %s" % (100 * wg, prev - t0, ", ".join(line)), flush=True)

# ---- input-gradient kernels (s2t_rowblock_dgrad): QKV (K = 768) and pointwise conv 1 (K = 512) with the LayerNorm backward,
# a plain K = 256 projection
ws = torch.zeros(K.LN_REPLICAS * 2 * 256, device=DEV)
for name, Kd, ln in (("dgrad qkv", 768, True), ("dgrad pw1", 512, True), ("dgrad 256", 256, False)):
    dy = torch.randn(M, Kd, generator=g).bfloat16().to(DEV)
    wt = (torch.randn(256, Kd, generator=g) * 256 ** -0.5).bfloat16().to(DEV)
    dx = torch.empty(M, 256, dtype=torch.bfloat16, device=DEV)
    dres = torch.randn(M, 256, generator=g).bfloat16().to(DEV)
    dxd = torch.empty_like(dx)

    def dg():
        if ln:
            K.rowblock_dgrad(dy, wt, ln=dict(x=x, gamma=gam, mean=mean, rstd=rstd, ws=ws, dx=dx, dres=dres, lens=lens, T=250,
                                             dx_drop=dxd, drop=(0.1, seed, 9)))
        else:
            K.rowblock_dgrad(dy, wt, dxn=dx)

    mean.zero_(); rstd.fill_(1.0)
    t = timeit(dg)
    print("%-9s K=%4d  rowblock %.1f us (%.0f TF/s)" % (name, Kd, t, 2.0 * M * 256 * Kd / t / 1e6), flush=True)
