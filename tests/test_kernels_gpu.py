"""GPU parity of the non-GEMM HIP kernels against the CPU oracle (oracle/s2t_oracle.py) / plain fp32 maths.
Every test runs the f32 and the bf16 storage flavour; statistics are fp32 in both."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import s2t_oracle as O  # noqa: E402
from s2t_amd import kernels as K  # noqa: E402

DEV = "cuda"
DT = [torch.float32, torch.bfloat16]


def tol(dtype):
    return dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)


def rnd(shape, dtype, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def close(got, ref, dtype, mul=1.0):
    t = tol(dtype)
    np.testing.assert_allclose(got.detach().cpu().double().numpy(), ref.detach().double().numpy(), rtol=t["rtol"] * mul,
                               atol=t["atol"] * mul)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cols", [32, 256, 260, 512])
def test_layernorm_fwd_bwd(dtype, cols):
    g = torch.Generator().manual_seed(cols)
    B, T = 3, 11
    rows = B * T
    x = rnd((rows, cols), dtype, g)
    w = 1 + 0.1 * torch.randn(cols, generator=g)
    b = 0.1 * torch.randn(cols, generator=g)
    dy = rnd((rows, cols), dtype, g)
    lens = torch.tensor([11, 7, 1], dtype=torch.int32)
    for use_mask in (False, True):
        xd, y = x.to(DEV), torch.empty(rows, cols, dtype=dtype, device=DEV)
        mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
        ld = lens.to(DEV) if use_mask else None
        K.layernorm_fwd(xd, w.to(DEV), b.to(DEV), y, mean, rstd, rows, cols, 1e-5, ld, T)
        xr = x.float().clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref = O.layer_norm(xr, wr, br)
        if use_mask:
            m = (torch.arange(T)[None] >= lens[:, None]).reshape(-1)
            ref = ref.masked_fill(m[:, None], 0.0)
        close(y, ref, dtype)
        ref.backward(dy.float())
        dx = torch.empty_like(xd)
        dg, db = torch.ones(cols, device=DEV), torch.ones(cols, device=DEV)  # accumulate semantics
        K.layernorm_bwd(xd, w.to(DEV), dy.to(DEV), mean, rstd, dx, dg, db, rows, cols, ld, T)
        close(dx, xr.grad, dtype, 2)
        close(dg - 1, wr.grad, dtype, 8)
        close(db - 1, br.grad, dtype, 8)
        # fused residual-branch gradient: dx = LN-backward(dy) + dres
        dres = rnd((rows, cols), dtype, g)
        dx2 = torch.empty_like(xd)
        dg2, db2 = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
        K.layernorm_bwd(xd, w.to(DEV), dy.to(DEV), mean, rstd, dx2, dg2, db2, rows, cols, ld, T, dres.to(DEV))
        close(dx2, xr.grad + dres.float(), dtype, 2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Tq,Tk,causal,rel", [(50, 50, False, False), (9, 70, False, False), (13, 13, True, False),
                                              (50, 50, False, True), (300, 300, False, True)])
def test_attn_softmax_fwd_bwd(dtype, Tq, Tk, causal, rel):
    g = torch.Generator().manual_seed(Tq * 3 + Tk)
    Bz, H = 2, 2
    Z = Bz * H
    ldS = (Tk + 7) // 8 * 8
    ldB = (2 * Tq - 1 + 7) // 8 * 8
    S = torch.randn(Z, Tq, ldS, generator=g) * 3
    BD = torch.randn(Z, Tq, ldB, generator=g) * 3 if rel else None
    klen = torch.tensor([Tk, max(1, Tk - 5)], dtype=torch.int32)
    scale = 0.37
    P = torch.full((Z, Tq, ldS), 9.0, dtype=dtype, device=DEV)
    K.attn_softmax_fwd(S.to(DEV), ldS, BD.to(DEV) if rel else None, ldB, P, ldS, Z, H, Tq, Tk, scale, klen.to(DEV),
                       causal, rel)
    s = S[..., :Tk].clone().requires_grad_(True)
    bdr = None
    sc = s
    if rel:
        bdr = BD.clone().requires_grad_(True)
        idx = (Tq - 1) - torch.arange(Tq)[:, None] + torch.arange(Tk)[None, :]
        sc = s + torch.gather(bdr, 2, idx[None].expand(Z, Tq, Tk))
    sc = sc * scale
    kmask = torch.arange(Tk)[None, :] >= klen[:, None]  # (B, Tk)
    sc = sc.masked_fill(kmask.repeat_interleave(H, 0)[:, None, :], float("-inf"))
    if causal:
        sc = sc + torch.triu(torch.full((Tq, Tk), float("-inf")), 1)
    if rel:
        sc = sc.clamp(-1e8, 1e8)
    ref = torch.softmax(sc, -1)
    close(P[..., :Tk], ref, dtype)
    assert (P[..., Tk:] == 0).all()
    dP = torch.randn(Z, Tq, ldS, generator=g)
    ref.backward(dP[..., :Tk])
    dS = torch.empty(Z, Tq, ldS, dtype=dtype, device=DEV)
    dBD = torch.full((Z, Tq, ldB), 5.0, dtype=dtype, device=DEV) if rel else None
    # backward consumes the probabilities as stored (bf16-rounded in bf16 mode)
    K.attn_softmax_bwd(P, ldS, dP.to(DEV), ldS, dS, ldS, dBD, ldB, Z, H, Tq, Tk, scale)
    close(dS[..., :Tk], s.grad, dtype, 2)
    if rel:
        # dBD comes back head-major: row (h*B + b)*Tq + i
        hm = bdr.grad.view(Bz, H, Tq, ldB).transpose(0, 1).reshape(Z, Tq, ldB)
        close(dBD[..., : 2 * Tq - 1], hm[..., : 2 * Tq - 1], dtype, 2)


@pytest.mark.parametrize("dtype", DT)
def test_positions_mask_embedding(dtype):
    g = torch.Generator().manual_seed(3)
    B, T, d = 3, 9, 32
    x = rnd((B, T, d), dtype, g)
    lens = torch.tensor([9, 6, 4], dtype=torch.int32)
    tab = O.sinusoidal_table(T + 2, d)
    xd = x.to(DEV)
    K.add_positions(xd, tab.to(DEV), lens.to(DEV), B * T, T, d, 1.0, 2)
    mask = O.lengths_to_padding_mask(lens.long(), T)
    ref = x.float() + O.sinusoidal_positions(mask, d)
    close(xd, ref, dtype)
    xd = x.to(DEV)
    K.mask_rows(xd, lens.to(DEV), B * T, T, d)
    close(xd, x.float().masked_fill(mask[:, :, None], 0.0), dtype)
    # embedding fwd/bwd
    V = 17
    E = rnd((V, d), dtype, g)
    tok = torch.tensor([[2, 5, 6, 7, 1], [2, 9, 1, 1, 1]])
    pos = ((tok != 1).long().cumsum(1) * (tok != 1).long() + 1).int()
    tab2 = O.sinusoidal_table(8, d)
    out = torch.empty(2, 5, d, dtype=dtype, device=DEV)
    K.embedding_fwd(tok.to(DEV), pos.to(DEV), E.to(DEV), tab2.to(DEV), out, 10, d, math.sqrt(d))
    ref = math.sqrt(d) * E.float()[tok] + O.sinusoidal_positions(tok, d)
    close(out, ref, dtype, 4)
    dout = rnd((2, 5, d), dtype, g)
    dE = torch.zeros(V, d, device=DEV)
    K.embedding_bwd(tok.to(DEV), dout.to(DEV), dE, 10, d, math.sqrt(d), 1)
    refE = torch.zeros(V, d)
    refE.index_add_(0, tok.reshape(-1), dout.float().reshape(-1, d) * math.sqrt(d))
    refE[1] = 0
    close(dE, refE, torch.float32, 4)


@pytest.mark.parametrize("dtype", DT)
def test_glu_bwd_colsum_cast_axpy(dtype):
    g = torch.Generator().manual_seed(4)
    B, T, n = 2, 7, 24
    rows = B * T
    Z = rnd((rows, 2 * n), dtype, g)
    dY = rnd((rows, n), dtype, g)
    lens = torch.tensor([7, 3], dtype=torch.int32)
    dZ = torch.empty(rows, 2 * n, dtype=dtype, device=DEV)
    K.glu_bwd(Z.to(DEV), dY.to(DEV), dZ, rows, n, lens.to(DEV), T)
    zr = Z.float().clone().requires_grad_(True)
    y = O.glu_channels(zr)
    m = (torch.arange(T)[None] >= lens[:, None]).reshape(-1)
    y.backward(dY.float().masked_fill(m[:, None], 0.0))
    close(dZ, zr.grad, dtype)
    # colsum over a strided view, odd width
    M, N = 1000, 70
    dy = rnd((M, 72), dtype, g)
    db = torch.ones(N, device=DEV)
    K.colsum_accum(dy.to(DEV), 72, db, M, N)
    close(db - 1, dy.float()[:, :N].sum(0), torch.float32, 50 if dtype == torch.bfloat16 else 5)
    # the 16-byte fast path (bf16, widths 256..2048, strided rows, ragged last slice)
    for Nw, Mr, ldw in ((256, 16000, 256), (512, 4099, 520), (1024, 777, 1024), (2048, 513, 2048)):
        dyw = rnd((Mr, ldw), dtype, g)
        dbw = torch.full((Nw,), 2.0, device=DEV)
        K.colsum_accum(dyw.to(DEV), ldw, dbw, Mr, Nw)
        close(dbw - 2, dyw.float()[:, :Nw].sum(0), torch.float32, 400 if dtype == torch.bfloat16 else 20)
    src = torch.randn(1024, generator=g)
    dst = torch.empty(1024, dtype=torch.bfloat16, device=DEV)
    K.cast_f32_to_bf16(src.to(DEV), dst, 1024)
    assert torch.equal(dst.cpu(), src.to(torch.bfloat16))
    a, b = rnd((512,), dtype, g), rnd((512,), dtype, g)
    yv = torch.empty(512, dtype=dtype, device=DEV)
    K.axpy(a.to(DEV), b.to(DEV), yv, 0.5, 512)
    close(yv, a.float() + 0.5 * b.float(), dtype)


def test_adam_clip_matches_fairseq_formula():
    g = torch.Generator().manual_seed(5)
    n = 4096
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    m, v = torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = p.to(DEV), gr.to(DEV), m.to(DEV), v.to(DEV)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    hyper = torch.zeros(4, device=DEV)
    sumsq = torch.zeros(1, device=DEV)
    lr, b1, b2, eps, max_norm, mult = 2e-3, 0.9, 0.98, 1e-8, 10.0, 0.25
    pr, mr, vr = p.clone().double(), m.clone().double(), v.clone().double()
    for step in (1, 2, 3):
        sumsq.zero_()
        K.sumsq_accum(gd, n, sumsq)
        K.clip_coef(sumsq, max_norm, mult, hyper)
        bc1, bc2 = 1 - b1**step, 1 - b2**step
        hyper[0] = lr
        hyper[1] = lr * math.sqrt(bc2) / bc1
        K.adam_step(pd, gd, md, vd, shadow, n, b1, b2, eps, 0.0, hyper)
        # reference: trainer.py:729-741 (multiply, clip) + optim/adam.py:146-226
        ge = gr.double() * mult
        norm = ge.norm()
        ge = ge * min(1.0, max_norm / (norm + 1e-6))
        mr = b1 * mr + (1 - b1) * ge
        vr = b2 * vr + (1 - b2) * ge * ge
        pr = pr - (lr * math.sqrt(bc2) / bc1) * mr / (vr.sqrt() + eps)
        assert abs(hyper[3].item() - norm.item()) < 1e-3 * norm.item()
    np.testing.assert_allclose(pd.cpu().double().numpy(), pr.numpy(), rtol=1e-5, atol=1e-6)
    assert torch.equal(shadow.cpu(), pd.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Kw", [15, 31])
def test_dwconv_bn_act_fwd_bwd(dtype, Kw):
    g = torch.Generator().manual_seed(Kw)
    B, T, C = 3, 45, 32
    lens = torch.tensor([45, 30, 8], dtype=torch.int32)
    x = rnd((B, T, C), dtype, g)
    w = torch.randn(C, Kw, generator=g) * 0.3
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    rm, rv = 0.1 * torch.randn(C, generator=g), 1 + 0.2 * torch.rand(C, generator=g)
    dOut = rnd((B, T, C), dtype, g)
    for act in ("relu", "swish"):
        # ---- training path: conv (+stats) -> finalize -> bn_act
        xd = x.to(DEV)
        D = torch.empty_like(xd)
        stats = torch.full((K.dwconv_stat_partials(B, T), 2, C), float("nan"), device=DEV)  # every row must be written
        K.dwconv_fwd(xd, w.to(DEV), D, B, T, C, Kw, stats=stats)
        scale, shift, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
        rmd, rvd = rm.to(DEV), rv.to(DEV)
        K.bn_finalize(stats, B * T, gamma.to(DEV), beta.to(DEV), rmd, rvd, 0.1, 1e-5, True, scale, shift, mean, rstd, C)
        out = torch.empty_like(xd)
        K.bn_act_fwd(D, out, scale, shift, act, B * T, C, lens.to(DEV), T)
        xr = x.float().clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        gr_, br_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        conv = torch.nn.functional.conv1d(xr.transpose(1, 2), wr[:, None, :], None, padding=(Kw - 1) // 2, groups=C).transpose(1, 2)
        mu = conv.mean((0, 1))
        var = ((conv - mu) ** 2).mean((0, 1))
        u = (conv - mu) / torch.sqrt(var + 1e-5) * gr_ + br_
        pm = O.lengths_to_padding_mask(lens.long(), T)
        ref = O.activation(act, u).masked_fill(pm[:, :, None], 0.0)
        close(D, conv, dtype, 2)
        close(out, ref, dtype, 4)
        n = B * T
        close(rmd, 0.9 * rm + 0.1 * mu, torch.float32, 200 if dtype == torch.bfloat16 else 5)
        close(rvd, 0.9 * rv + 0.1 * var * n / (n - 1), torch.float32, 200 if dtype == torch.bfloat16 else 5)
        # ---- backward
        ref.backward(dOut.float())
        dD = torch.empty_like(xd)
        sums = torch.full((2 * C,), float("nan"), device=DEV)
        K.bn_act_bwd(D, dOut.to(DEV), dD, scale, shift, mean, rstd, sums, n, act, n, C, lens.to(DEV), T)
        dG = torch.empty_like(xd)
        K.dwconv_fwd(dD, w.to(DEV), dG, B, T, C, Kw, flip=True)
        dw = torch.zeros(C, Kw, device=DEV)
        K.dwconv_bwd_weight(xd, dD, dw, B, T, C, Kw)
        m = 1 if dtype == torch.float32 else 1
        close(dG, xr.grad, dtype, 8 * m)
        # bf16: ~135 products of bf16-rounded dD and x per entry, |dw| up to ~10 -> absolute error ~2e-2
        close(dw, wr.grad, torch.float32, 2500 if dtype == torch.bfloat16 else 20)
        close(sums[:C], br_.grad, torch.float32, 2500 if dtype == torch.bfloat16 else 20)
        close(sums[C:], gr_.grad, torch.float32, 2500 if dtype == torch.bfloat16 else 20)
        # ---- eval path: BN folded into the conv epilogue
        K.bn_finalize(None, 0, gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), 0.1, 1e-5, False, scale, shift, None, None, C)
        out2 = torch.empty_like(xd)
        K.dwconv_fwd(xd, w.to(DEV), out2, B, T, C, Kw, scale=scale, shift=shift, act=act, lens=lens.to(DEV))
        ue = (conv.detach() - rm) / torch.sqrt(rv + 1e-5) * gamma + beta
        close(out2, O.activation(act, ue).masked_fill(pm[:, :, None], 0.0), dtype, 4)


@pytest.mark.parametrize("B,T,C,Kw,act", [(3, 45, 32, 15, "swish"), (2, 250, 256, 15, "swish"), (4, 70, 260, 31, "relu")])
def test_conv_bwd_fused_matches_the_four_kernel_chain(B, T, C, Kw, act):
    """s2t_conv_bwd_fused (bf16) against the chain it replaces on the same inputs: BatchNorm apply pass -> flipped depthwise
    conv -> GLU backward, and the depthwise weight gradient (modules/convolution.py:92-104 backward).  dD and dG are rounded
    to bf16 at the same points, so dZ agrees to a bf16 ulp or two and the weight gradient to fp32 summation order."""
    g = torch.Generator().manual_seed(B * T + C)
    bf = torch.bfloat16
    lens = torch.tensor([T, max(1, T * 2 // 3), 8, T][:B], dtype=torch.int32).to(DEV)
    G = rnd((B, T, C), bf, g).to(DEV)                       # GLU output the forward convolved
    Z = rnd((B * T, 2 * C), bf, g).to(DEV)                  # value | gate
    w = (torch.randn(C, Kw, generator=g) * 0.3).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    dA = rnd((B, T, C), bf, g).to(DEV)
    n = B * T
    D = torch.empty_like(G)
    stats = torch.empty(K.dwconv_stat_partials(B, T), 2, C, device=DEV)
    K.dwconv_fwd(G, w, D, B, T, C, Kw, stats=stats)
    scale, shift, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
    K.bn_finalize(stats, n, gamma, beta, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), 0.1, 1e-5, True, scale, shift,
                  mean, rstd, C)
    # the chain
    dD = torch.empty_like(G)
    sums = torch.empty(2 * C, device=DEV)
    K.bn_act_bwd(D, dA, dD, scale, shift, mean, rstd, sums, n, act, n, C, lens, T)
    dG = torch.empty_like(G)
    K.dwconv_fwd(dD, w, dG, B, T, C, Kw, flip=True)
    dw_ref = torch.full((C, Kw), 0.5, device=DEV)
    K.dwconv_bwd_weight(G, dD, dw_ref, B, T, C, Kw)
    dZ_ref = torch.empty_like(Z)
    K.glu_bwd(Z, dG.view(n, C), dZ_ref, n, C)
    # the fused launch (reduce + fold first, without the apply pass)
    sums2 = torch.empty(2 * C, device=DEV)
    K.bn_act_bwd(D, dA, None, scale, shift, mean, rstd, sums2, n, act, n, C, lens, T)
    assert torch.equal(sums, sums2)
    dZ = torch.full_like(Z, 3.0)
    dw = torch.full((C, Kw), 0.5, device=DEV)
    K.conv_bwd_fused(D.view(n, C), dA.view(n, C), G.view(n, C), Z, w, scale, shift, mean, rstd, sums2, n, act, lens, dZ, dw, B,
                     T, C, Kw)
    torch.cuda.synchronize()
    err = (dZ.float() - dZ_ref.float()).abs().max() / dZ_ref.float().abs().max()
    assert err < 1.6e-2, float(err)
    assert (dZ.float() - dZ_ref.float()).norm() / dZ_ref.float().norm() < 4e-3
    assert (dw - dw_ref).abs().max() <= 2e-3 * dw_ref.abs().max() + 1e-4
    # deferred fold: the partial rows of two modules folded by one s2t_rows_fold_add launch give the same bits
    dws = [torch.full((C, Kw), 0.5, device=DEV), torch.full((C, Kw), -1.25, device=DEV)]
    parts = []
    for k in range(2):
        ws, rows = K.conv_bwd_fused(D.view(n, C), dA.view(n, C), G.view(n, C), Z, w, scale, shift, mean, rstd, sums2, n, act, lens,
                                    torch.empty_like(Z), dws[k], B, T, C, Kw, defer_slot=k)
        parts.append(ws)
    assert torch.equal(dws[0], torch.full((C, Kw), 0.5, device=DEV))  # untouched until the fold
    K.rows_fold_add(parts, dws, rows, C * Kw)
    torch.cuda.synchronize()
    assert torch.equal(dws[0], dw)
    assert torch.equal(dws[1] + 1.75, dw) or (dws[1] + 1.75 - dw).abs().max() <= 1e-6 * dw.abs().max()


@pytest.mark.parametrize("dtype", DT)
def test_ctc_greedy_kernels(dtype):
    g = torch.Generator().manual_seed(6)
    B, T, V = 4, 1100, 300  # T > 1024 exercises the chunked scan
    logits = rnd((B, T, V), dtype, g, 2.0)
    # long runs + blanks so that collapse matters
    runs = torch.randint(0, 6, (B, T // 4 + 1), generator=g).repeat_interleave(4, 1)[:, :T]
    logits.scatter_(2, runs[:, :, None], 20.0)
    lens = torch.tensor([1100, 900, 513, 1], dtype=torch.int32)
    ld = logits.to(DEV)
    idx = torch.empty(B * T, dtype=torch.int32, device=DEV)
    top = torch.empty(B * T, device=DEV)
    K.argmax_lse(ld, V, B * T, V, idx, top, None)
    toks = torch.zeros(B, T, dtype=torch.int64, device=DEV)
    olen = torch.zeros(B, dtype=torch.int32, device=DEV)
    osc = torch.zeros(B, device=DEV)
    K.ctc_collapse(idx, top, lens.to(DEV), B, T, 0, toks, olen, osc)
    hyps, scores = O.ctc_greedy(logits.float().transpose(0, 1), O.lengths_to_padding_mask(lens.long(), T))
    for b in range(B):
        n = int(olen[b])
        assert n == len(hyps[b])
        assert toks[b, :n].cpu().tolist() == hyps[b].tolist()  # bit-exact ids
    np.testing.assert_allclose(osc.cpu().numpy(), scores.numpy(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("dtype", DT)
def test_argmax_ties_pick_lowest_index(dtype):
    x = torch.zeros(3, 500, dtype=dtype)
    x[0, [7, 300]] = 5.0
    x[1, [499, 2, 256]] = 1.0
    idx = torch.empty(3, dtype=torch.int32, device=DEV)
    K.argmax_lse(x.to(DEV), 500, 3, 500, idx, None, None)
    assert idx.cpu().tolist() == [7, 2, 0]


@pytest.mark.parametrize("dtype", DT)
def test_label_smoothed_ce(dtype):
    g = torch.Generator().manual_seed(7)
    rows, V = 23, 1000
    logits = rnd((rows, V), dtype, g, 2.0)
    tgt = torch.randint(2, V, (rows,), generator=g)
    tgt[[3, 9, 22]] = 1
    sums = torch.zeros(4, device=DEV)
    dl = torch.empty(rows, V, dtype=dtype, device=DEV)
    K.ls_cross_entropy(logits.to(DEV), V, rows, V, tgt.to(DEV), 1, 0.1, dl, V, sums)
    lr = logits.float().clone().requires_grad_(True)
    loss, nll = O.label_smoothed_nll(lr, tgt, 0.1)
    loss.backward()
    nc, tot = O.ce_accuracy(logits.float(), tgt)
    s = sums.cpu()
    assert abs(s[0].item() - loss.item()) < 2e-4 * abs(loss.item())
    assert abs(s[1].item() - nll.item()) < 2e-4 * abs(nll.item())
    assert int(s[2]) == nc and int(s[3]) == tot
    close(dl, lr.grad, dtype, 1 if dtype == torch.float32 else 0.5)


def _ls_ce(logits, tgt, eps, pad=1):
    rows, V = logits.shape
    sums = torch.zeros(4, device=DEV)
    K.ls_cross_entropy(logits.to(DEV).contiguous(), V, rows, V, tgt.to(DEV), pad, eps, None, 0, sums)
    return sums.cpu()


def _ls_fixture():
    """The shape of the reference's own fixture (/root/reference/tests/test_label_smoothing.py:28-60): a batch of two target
    rows of three positions over a small vocabulary, the second utterance one token shorter (its last position is padding,
    index 1), fixed per-position distributions."""
    g = torch.Generator().manual_seed(11)
    V = 9
    probs = torch.rand(2, 3, V, generator=g) + 0.05
    probs = probs / probs.sum(-1, keepdim=True)
    logits = probs.log() + 0.7  # any per-row shift: log_softmax removes it
    target = torch.tensor([[4, 5, 2], [6, 2, 1]])
    return logits, target


def test_label_smoothing_nll_is_the_plain_cross_entropy():
    """tests/test_label_smoothing.py:62-75 (test_nll_loss): the nll the smoothed criterion reports is CrossEntropy's loss."""
    logits, target = _ls_fixture()
    s = _ls_ce(logits.view(-1, logits.shape[-1]), target.view(-1), 0.1)
    lp = torch.log_softmax(logits.view(-1, logits.shape[-1]), -1)
    plain = torch.nn.functional.nll_loss(lp, target.view(-1), ignore_index=1, reduction="sum")
    assert abs(float(s[1]) - float(plain)) < 1e-5
    assert int(s[3]) == 5


def test_label_smoothing_padding_adds_nothing():
    """:77-95 (test_padding): the loss of the padded batch is the sum of the losses of its utterances run alone, unpadded."""
    logits, target = _ls_fixture()
    V = logits.shape[-1]
    both = _ls_ce(logits.view(-1, V), target.view(-1), 0.1)
    one = _ls_ce(logits[0], target[0], 0.1)
    two = _ls_ce(logits[1, :2], target[1, :2], 0.1)
    assert abs(float(both[0]) - float(one[0]) - float(two[0])) < 1e-5
    assert abs(float(both[1]) - float(one[1]) - float(two[1])) < 1e-5


def test_label_smoothing_reduction_is_the_sum_over_positions():
    """:97-102 (test_reduction): the reduced loss equals the sum of the unreduced per-position losses (each position run as a
    launch of its own here: the kernel only has the reduced form)."""
    logits, target = _ls_fixture()
    V = logits.shape[-1]
    total = _ls_ce(logits.view(-1, V), target.view(-1), 0.1)
    parts = sum(float(_ls_ce(logits.view(-1, V)[i:i + 1], target.view(-1)[i:i + 1], 0.1)[0]) for i in range(6))
    assert abs(float(total[0]) - parts) < 1e-5


def test_label_smoothing_zero_eps_is_nll():
    """:104-116 (test_zero_eps): without smoothing the smoothed loss IS the nll."""
    logits, target = _ls_fixture()
    s = _ls_ce(logits.view(-1, logits.shape[-1]), target.view(-1), 0.0)
    assert abs(float(s[0]) - float(s[1])) < 1e-6
    lp = torch.log_softmax(logits.view(-1, logits.shape[-1]), -1)
    plain = torch.nn.functional.nll_loss(lp, target.view(-1), ignore_index=1, reduction="sum")
    assert abs(float(s[0]) - float(plain)) < 1e-5


@pytest.mark.parametrize("longest", [15, 63, 70])
@pytest.mark.parametrize("dtype", DT)
def test_ctc_loss_fwd_bwd(dtype, longest):
    """longest <= 63 labels (<= 127 states) runs the single-wave recursion, 70 the one-thread-per-state workgroup."""
    g = torch.Generator().manual_seed(8)
    B, T, V = 5, 40 if longest == 15 else 150, 23
    logits = rnd((B, T, V), dtype, g, 1.5)
    tg = [torch.tensor([4, 4, 5, 9]), torch.tensor([3]), torch.tensor([], dtype=torch.long),
          torch.tensor([7, 8, 7, 8, 7, 8, 7, 8, 7, 8, 7, 8]), torch.randint(1, V, (longest,), generator=g)]
    in_lens = torch.tensor([40, 33, 10, 11, T], dtype=torch.int32)  # utt 3: 12 labels in 11 frames -> infeasible
    S = max(len(t) for t in tg)
    tmat = torch.zeros(B, S, dtype=torch.int64)
    for b, t in enumerate(tg):
        tmat[b, : len(t)] = t
    tl = torch.tensor([len(t) for t in tg], dtype=torch.int32)
    Lmax = 2 * S + 1
    ld = logits.to(DEV)
    lse = torch.empty(B * T, device=DEV)
    K.argmax_lse(ld, V, B * T, V, None, None, lse)
    alpha = torch.full((B, T, Lmax), float("nan"), device=DEV)
    beta = torch.full((B, T, Lmax), float("nan"), device=DEV)
    nll = torch.empty(B, device=DEV)
    K.ctc_loss_fwd(ld, V, B, T, V, lse, tmat.to(DEV), S, tl.to(DEV), in_lens.to(DEV), 0, alpha, beta, Lmax, nll)
    lr = logits.float().clone().requires_grad_(True)
    lp = torch.log_softmax(lr.transpose(0, 1), -1)
    ref = O.ctc_nll(lp, tg, in_lens.long())
    got = nll.cpu()
    assert torch.isinf(got[3])  # raw nll; zero_infinity is applied by the caller / gradient kernel
    got = torch.where(torch.isinf(got), torch.zeros_like(got), got)
    np.testing.assert_allclose(got.numpy(), ref.detach().numpy(), rtol=2e-4, atol=2e-3 if dtype == torch.float32 else 5e-2)
    (0.3 * ref.sum()).backward()
    grad = torch.empty(B, T, V, dtype=dtype, device=DEV)
    K.ctc_loss_bwd(ld, V, B, T, V, lse, tmat.to(DEV), S, tl.to(DEV), in_lens.to(DEV), 0, alpha, beta, Lmax, nll, 0.3, grad, V)
    close(grad, lr.grad, dtype, 5 if dtype == torch.float32 else 1)


@pytest.mark.parametrize("dtype", DT)
def test_ctc_compress_plan_and_rows(dtype):
    """s2t_ctc_compress_plan + s2t_compress_rows (s2t_transformer.py:1948-1986) against boolean indexing in torch:
    ragged lengths, an utterance that keeps nothing, T > 256 (several scan rounds), forward gather and backward scatter."""
    from s2t_amd import functional as Fn
    g = torch.Generator().manual_seed(12)
    B, T, V, d = 5, 300, 17, 64
    lens = torch.tensor([300, 257, 256, 31, 7], dtype=torch.int32)
    logits = rnd((B * T, V), dtype, g, scale=2.0)
    logits[:, 0] += 1.5
    logits.view(B, T, V)[4, :, 0] = 30.0  # utterance 4 is all blank
    x = rnd((B * T, d), dtype, g)
    thr = 0.4
    ld = logits.to(DEV)
    src, new_lens = Fn.ctc_compress_plan(ld, lens.to(DEV), B, T, 0, thr)
    pb = torch.softmax(ld.float().view(B, T, V), -1)[:, :, 0].cpu()  # the same rounded logits the kernel saw
    margin = (pb - thr).abs()
    keep = (pb < thr) & (torch.arange(T)[None, :] < lens[:, None])
    assert float(margin[torch.arange(T)[None, :] < lens[:, None]].min()) > 1e-4  # no decision sits on the threshold
    assert new_lens.cpu().tolist() == keep.sum(1).tolist() and int(new_lens[4]) == 0
    for b in range(B):
        assert src[b, : int(new_lens[b])].cpu().tolist() == torch.nonzero(keep[b]).view(-1).tolist()
    Tn = int(new_lens.max())
    xd = x.to(DEV).requires_grad_(True)
    y = Fn.CompressRowsFn.apply(xd, src, new_lens, B, T, Tn)
    ref = torch.zeros(B, Tn, d, dtype=x.dtype)
    for b in range(B):
        ref[b, : int(keep[b].sum())] = x.view(B, T, d)[b][keep[b]]
    assert torch.equal(y.detach().cpu().view(B, Tn, d), ref)  # pure data movement: bit-exact
    dy = rnd((B * Tn, d), dtype, g)
    y.backward(dy.to(DEV))
    dref = torch.zeros(B, T, d, dtype=x.dtype)
    for b in range(B):
        dref[b][keep[b]] = dy.view(B, Tn, d)[b, : int(keep[b].sum())]
    assert torch.equal(xd.grad.cpu().view(B, T, d), dref)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("r", [1, 2, 4])
def test_dwpool_bn_act_fwd_bwd(dtype, r):
    """s2t_dwpool_fwd/_bwd + BatchNorm + Swish (downsample_convolution.py:97-106) against torch: ragged tail (T % r != 0),
    statistics over all rows, running-stat update, gradients of x, depthwise weight / bias and BatchNorm affine."""
    from s2t_amd import functional as Fn
    g = torch.Generator().manual_seed(20 + r)
    B, T, C = 3, 70 + (1 if r > 1 else 0), 32
    x = rnd((B * T, C), dtype, g)
    w = torch.nn.Parameter((torch.randn(C, 1, r, generator=g) * 0.5).to(DEV))
    b = torch.nn.Parameter((torch.randn(C, generator=g) * 0.1).to(DEV))
    gamma = torch.nn.Parameter((1 + 0.1 * torch.randn(C, generator=g)).to(DEV))
    beta = torch.nn.Parameter((0.1 * torch.randn(C, generator=g)).to(DEV))
    for p in (w, b, gamma, beta):
        p.grad = torch.zeros_like(p)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    xd = x.to(DEV).requires_grad_(True)
    prm = {"dw_w": w, "dw_b": b, "bn_w": gamma, "bn_b": beta}
    a = Fn.pool_bn_act(xd, prm, {"running_mean": rm, "running_var": rv}, "swish", B, T, r, True)
    To = T // r
    dA = rnd((B * To, C), dtype, g)
    a.backward(dA.to(DEV))
    xr = x.float().view(B, T, C).clone().requires_grad_(True)
    wr, br = w.detach().cpu().clone().requires_grad_(True), b.detach().cpu().clone().requires_grad_(True)
    gr_, ber = gamma.detach().cpu().clone().requires_grad_(True), beta.detach().cpu().clone().requires_grad_(True)
    D = torch.nn.functional.conv1d(xr.transpose(1, 2), wr, br, stride=r, groups=C).transpose(1, 2)
    mu, var = D.mean((0, 1)), ((D - D.mean((0, 1))) ** 2).mean((0, 1))
    u = (D - mu) / torch.sqrt(var + 1e-5) * gr_ + ber
    ref = u * torch.sigmoid(u)
    ref.backward(dA.float().view(B, To, C))
    m = 4 if dtype == torch.bfloat16 else 1
    close(a.view(B, To, C), ref, dtype, 4)
    n = B * To
    close(rm, 0.1 * mu, torch.float32, 200 if dtype == torch.bfloat16 else 5)
    close(rv, 0.9 + 0.1 * var * n / (n - 1), torch.float32, 200 if dtype == torch.bfloat16 else 5)
    close(xd.grad.view(B, T, C), xr.grad, dtype, 10 * m)
    # BatchNorm removes a per-channel scale / shift of its input, so dw and db are differences of large, nearly
    # cancelling sums (for r = 1 they are mathematically zero up to eps): compare on the scale of the gamma gradient
    gs = float(gr_.grad.abs().max())
    for got, want in ((w.grad, wr.grad), (b.grad, br.grad), (gamma.grad, gr_.grad), (beta.grad, ber.grad)):
        err = float((got.detach().cpu() - want).abs().max())
        assert err <= (0.2 if dtype == torch.bfloat16 else 2e-3) * max(gs, float(want.abs().max())), (err, gs)  # bf16: dD is stored rounded


def test_add_colsum2_relpos_glue():
    """s2t_add_colsum2: a += b in place with the column sums of the old a and of b (strided a, ragged last slice)."""
    g = torch.Generator().manual_seed(33)
    rows, n = 16003, 256
    a = rnd((rows, 3 * n), torch.bfloat16, g)
    b = rnd((rows, n), torch.bfloat16, g)
    ad, bd = a.to(DEV), b.to(DEV)
    du, dv = torch.full((n,), 1.0, device=DEV), torch.full((n,), -2.0, device=DEV)
    K.add_colsum2(ad, 3 * n, bd, n, du, dv, rows, n)
    close(du - 1, a[:, :n].float().sum(0), torch.float32, 400)
    close(dv + 2, b.float().sum(0), torch.float32, 400)
    want = (a[:, :n].float() + b.float()).to(torch.bfloat16)
    assert torch.equal(ad[:, :n].cpu(), want)
    assert torch.equal(ad[:, n:].cpu(), a[:, n:])  # the k / v slices are untouched


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,V,ld", [(37, 10000, 10000), (5, 2048, 2056), (9, 10240, 10240), (6, 1000, 1000), (4, 37, 40), (3, 2050, 2056),
                                       (2, 12000, 12000)])
def test_row_softmax_fwd(dtype, rows, V, ld):
    """softmax(x / tau) over a wide vocabulary (the PAE distribution, modules/speech_to_text/adapter.py:214-217): the
    register-resident bf16 form (V % 8 == 0, V <= 10240) and the three-pass form against torch."""
    g = torch.Generator().manual_seed(V + rows)
    x = torch.full((rows, ld), float("nan"), dtype=dtype)
    x[:, :V] = rnd((rows, V), dtype, g) * 4
    xd = x.to(DEV)
    for tau in (1.0, 0.5):
        p = torch.full((rows, ld), 7.0, dtype=dtype, device=DEV)
        K.row_softmax_fwd(xd, ld, p, ld, rows, V, 1.0 / tau)
        ref = torch.softmax(x[:, :V].double() / tau, -1)
        np.testing.assert_allclose(p[:, :V].double().cpu().numpy(), ref.numpy(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-5,
                                   atol=1e-6)
        assert (p[:, V:] == 7.0).all()
        assert abs(p[:, :V].double().sum(-1).cpu() - 1).max() < (2e-2 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,V,ld", [(37, 10000, 10000), (5, 2048, 2056), (6, 1000, 1000), (4, 37, 40), (3, 2050, 2056), (2, 12000, 12000)])
def test_row_softmax_bwd(dtype, rows, V, ld):
    """dx = P * (dP - sum_j P dP) / tau (the PAE softmax's backward): both the register-resident bf16 form and the two-pass
    form against autograd through torch.softmax."""
    g = torch.Generator().manual_seed(V * 3 + rows)
    x = (rnd((rows, V), torch.float32, g) * 3).requires_grad_(True)
    tau = 0.7
    P = torch.softmax(x / tau, -1)
    dP = rnd((rows, V), torch.float32, g)
    Pq, dPq = P.detach().to(dtype), dP.to(dtype)
    (torch.softmax(x / tau, -1) * dPq.float()).sum().backward()
    # reference on the ROUNDED operands the kernel sees
    ref = Pq.double() * (dPq.double() - (Pq.double() * dPq.double()).sum(-1, keepdim=True)) / tau
    pd = torch.zeros(rows, ld, dtype=dtype, device=DEV); pd[:, :V] = Pq.to(DEV)
    dd = torch.zeros(rows, ld, dtype=dtype, device=DEV); dd[:, :V] = dPq.to(DEV)
    dx = torch.full((rows, ld), 7.0, dtype=dtype, device=DEV)
    K.row_softmax_bwd(pd, ld, dd, ld, dx, ld, rows, V, 1.0 / tau)
    got = dx[:, :V].double().cpu()
    scale = ref.abs().max().item()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-5,
                               atol=(1e-2 if dtype == torch.bfloat16 else 1e-6) * scale)
    assert (dx[:, V:] == 7.0).all()
