"""GPU parity of the MFMA GEMM (s2t_gemm) against a float64 CPU matmul (the oracle of a GEMM).

Covers all four operand layouts, both dtypes, ragged M/N/K (not multiples of the tile or of the
16-byte chunk), two-level batching, and every epilogue stage."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from s2t_amd import kernels as ops  # noqa: E402


def _mk(shape, dtype, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def _pad_ld(n, dtype):
    epb = 4 if dtype == torch.float32 else 8
    return (n + epb - 1) // epb * epb


def _tol(dtype, K):
    # bf16 inputs are exact in the reference product (cast first); only the fp32 accumulation order
    # differs; outputs rounded to bf16 add 2^-8 relative
    return (2e-5, 2e-5) if dtype == torch.float32 else (1e-2, 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 130, 100), (16, 10, 8), (300, 250, 64), (64, 1000, 256), (250, 64, 250)])
def test_layouts(dtype, akm, bkm, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    dev = "cuda"
    # storage with padded leading dims (row starts stay 16-byte aligned)
    if akm:
        lda = _pad_ld(M, dtype); A_st = torch.zeros(K, lda, dtype=dtype); A_log = _mk((M, K), dtype, g); A_st[:, :M] = A_log.t()
    else:
        lda = _pad_ld(K, dtype); A_st = torch.zeros(M, lda, dtype=dtype); A_log = _mk((M, K), dtype, g); A_st[:, :K] = A_log
    if bkm:
        ldb = _pad_ld(N, dtype); B_st = torch.zeros(K, ldb, dtype=dtype); B_log = _mk((K, N), dtype, g); B_st[:, :N] = B_log
    else:
        ldb = _pad_ld(K, dtype); B_st = torch.zeros(N, ldb, dtype=dtype); B_log = _mk((K, N), dtype, g); B_st[:, :K] = B_log.t()
    # poison the padding so that a kernel reading it shows up
    if akm: A_st[:, M:] = float("nan")
    else: A_st[:, K:] = float("nan")
    if bkm: B_st[:, N:] = float("nan")
    else: B_st[:, K:] = float("nan")
    ref = A_log.double() @ B_log.double()
    ldc = N + 3
    out = torch.full((M, ldc), 7.0, dtype=dtype, device=dev)
    ops.gemm(A_st.to(dev), B_st.to(dev), out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, a_kmajor=akm, b_kmajor=bkm)
    torch.cuda.synchronize()
    got = out.cpu().double()
    rtol, atol = _tol(dtype, K)
    scale = ref.abs().max().item()
    np.testing.assert_allclose(got[:, :N].numpy(), ref.numpy(), rtol=rtol, atol=atol * scale)
    assert (got[:, N:] == 7.0).all(), "wrote outside the N columns"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_epilogue_bias_act_residual_mask(dtype):
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    B_, T, K, N = 3, 50, 72, 136
    M = B_ * T
    A = _mk((M, K), dtype, g); W = _mk((N, K), dtype, g, 0.2); bias = _mk((N,), torch.float32, g)
    R = _mk((M, N), dtype, g)
    lens = torch.tensor([50, 31, 7], dtype=torch.int32)
    for act in ("relu", "swish", None):
        out = torch.empty(M, N, dtype=dtype, device=dev)
        pre = torch.empty(M, N, dtype=dtype, device=dev)
        ops.gemm(A.to(dev), W.to(dev), out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias.to(dev), act=act, alpha=0.5,
                 residual=R.to(dev), ldr=N, preact=pre if act else None, ldp=N, row_lens=lens.to(dev), row_T=T)
        z = A.double() @ W.double().t() + bias.double()
        a = {"relu": torch.relu, "swish": lambda v: v * torch.sigmoid(v), None: lambda v: v}[act](z)
        mask = (torch.arange(T)[None, :] >= lens[:, None]).reshape(-1)
        br = 0.5 * a
        br[mask] = 0  # the branch is masked, the residual is not
        ref = R.double() + br
        rtol, atol = _tol(dtype, K)
        np.testing.assert_allclose(out.cpu().double().numpy(), ref.numpy(), rtol=rtol, atol=atol * 4)
        if act:
            np.testing.assert_allclose(pre.cpu().double().numpy(), z.numpy(), rtol=rtol, atol=atol * 4)
    # dact: out = (A @ W^T) * act'(Z)
    Z = _mk((M, N), dtype, g)
    for act in ("relu", "swish"):
        out = torch.empty(M, N, dtype=dtype, device=dev)
        ops.gemm(A.to(dev), W.to(dev), out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, dact_z=Z.to(dev), ldz=N, dact=act)
        zd = Z.double()
        if act == "relu":
            d = (zd > 0).double()
        else:
            s = torch.sigmoid(zd); d = s * (1 + zd * (1 - s))
        ref = (A.double() @ W.double().t()) * d
        rtol, atol = _tol(dtype, K)
        np.testing.assert_allclose(out.cpu().double().numpy(), ref.numpy(), rtol=rtol, atol=atol * 4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_glu_epilogue_and_conv_rows(dtype):
    """Conv1d(k=5, stride 2, pad 2) + GLU as an overlapping-row GEMM over a zero-padded (B, T+4, C) buffer —
    the formulation the subsampler uses (reference: modules/speech_to_text/subsampling.py:145-159)."""
    g = torch.Generator().manual_seed(9)
    dev = "cuda"
    Bz, T, Cin, Cout = 3, 37, 16, 80  # Cout must be even; out channels = 40
    x = _mk((Bz, T, Cin), dtype, g)
    w = _mk((Cout, Cin, 5), dtype, g, 0.2)
    b = _mk((Cout,), torch.float32, g)
    Tout = (T - 1) // 2 + 1
    Tp = 2 * Tout + 4
    xp = torch.zeros(Bz, Tp, Cin, dtype=dtype)
    xp[:, 2:2 + T] = x
    wk = w.permute(0, 2, 1).contiguous().view(Cout, 5 * Cin)  # [Cout][k*Cin + c]
    out = torch.empty(Bz, Tout, Cout // 2, dtype=dtype, device=dev)
    pre = torch.empty(Bz, Tout, Cout, dtype=dtype, device=dev)
    lens = torch.tensor([Tout, Tout - 3, 5], dtype=torch.int32)
    ops.gemm(xp.to(dev), wk.to(dev), out, M=Tout, N=Cout, K=5 * Cin, lda=2 * Cin, ldb=5 * Cin, ldc=Cout // 2,
             batch=Bz, a_s=(Tp * Cin, 0), c_s=(Tout * (Cout // 2), 0), bias=b.to(dev), act="glu",
             preact=pre, ldp=Cout, p_s=(Tout * Cout, 0), row_lens=lens.to(dev), row_T=Tout)
    y = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), b.double(), stride=2, padding=2)  # (B, Cout, Tout)
    a, gt = y.chunk(2, dim=1)
    ref = (a * torch.sigmoid(gt)).transpose(1, 2).clone()
    for i, l in enumerate(lens.tolist()):
        ref[i, l:] = 0
    rtol, atol = _tol(dtype, 5 * Cin)
    np.testing.assert_allclose(out.cpu().double().numpy(), ref.numpy(), rtol=rtol, atol=atol * 4)
    np.testing.assert_allclose(pre.cpu().double().numpy(), y.transpose(1, 2).numpy(), rtol=rtol, atol=atol * 4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_two_level_and_splitk(dtype):
    g = torch.Generator().manual_seed(11)
    dev = "cuda"
    Bz, H, T, dk = 3, 2, 50, 16
    d = H * dk
    q = _mk((Bz, T, d), dtype, g); k = _mk((Bz, T, d), dtype, g)
    ldS = _pad_ld(T, dtype)
    S = torch.zeros(Bz, H, T, ldS, dtype=dtype, device=dev)
    # S[b,h] = q[b,:,h] @ k[b,:,h]^T  (z = b*H + h)
    ops.gemm(q.to(dev), k.to(dev), S, M=T, N=T, K=dk, lda=d, ldb=d, ldc=ldS, batch=Bz * H, zdiv=H,
             a_s=(T * d, dk), b_s=(T * d, dk), c_s=(H * T * ldS, T * ldS), alpha=0.25)
    qh = q.double().view(Bz, T, H, dk).transpose(1, 2); kh = k.double().view(Bz, T, H, dk).transpose(1, 2)
    ref = 0.25 * qh @ kh.transpose(-1, -2)
    rtol, atol = _tol(dtype, dk)
    np.testing.assert_allclose(S.cpu().double()[..., :T].numpy(), ref.numpy(), rtol=rtol, atol=atol * 4)
    # split-K wgrad: dW[N,K] += dY^T[N,M] X[M,K] accumulated into fp32
    M, N, K = 1000, 72, 40
    dY = _mk((M, N), dtype, g); X = _mk((M, K), dtype, g)
    dW = torch.ones(N, K, dtype=torch.float32, device=dev)
    ops.gemm(dY.to(dev), X.to(dev), dW, M=N, N=K, K=M, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, split_k=4)
    ref = 1.0 + dY.double().t() @ X.double()
    np.testing.assert_allclose(dW.cpu().double().numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_splitk_workspace_matches_atomics_and_reference(dtype):
    """Two-phase split-K (partials in a workspace + reduce kernel) vs the atomic path vs float64, incl. ragged tiles,
    alpha, the fused bias gradient and head-batched column blocks of one C."""
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    M, N, K = 2000, 296, 200  # dW[N=296, K=200] += dY^T X over M=2000 rows (bf16 rows must stay 16-byte aligned)
    dY = _mk((M, N), dtype, g); X = _mk((M, K), dtype, g)
    ref = 3.0 + 0.5 * (dY.double().t() @ X.double())
    refb = 1.0 + 0.5 * dY.double().sum(0)
    outs = []
    for use_ws in (True, False):
        dW = torch.full((N, K), 3.0, dtype=torch.float32, device=dev)
        db = torch.ones(N, dtype=torch.float32, device=dev)
        ops.gemm(dY.to(dev), X.to(dev), dW, M=N, N=K, K=M, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, split_k=7,
                 c_atomic=True, alpha=0.5, colsum_a=db, splitk_workspace=use_ws)
        torch.cuda.synchronize()
        np.testing.assert_allclose(dW.cpu().double().numpy(), ref.numpy(), rtol=1e-3, atol=1e-3 * ref.abs().max().item())
        np.testing.assert_allclose(db.cpu().double().numpy(), refb.numpy(), rtol=1e-3, atol=1e-3 * refb.abs().max().item())
        outs.append(dW.cpu())
    assert (outs[0] - outs[1]).abs().max() < 1e-3 * ref.abs().max().item()
    # head-batched: dp[n, h*dk + c] += sum_m A[h][m][n] * Q[m][h*dk + c]  (batch = H, one C, disjoint column blocks)
    H, dk, Mq, P = 3, 64, 900, 150
    A = _mk((H, Mq, 152), dtype, g); Q = _mk((Mq, H * dk), dtype, g)
    dp = torch.zeros(P, H * dk, dtype=torch.float32, device=dev)
    ops.gemm(A.to(dev), Q.to(dev), dp, M=P, N=dk, K=Mq, lda=152, ldb=H * dk, ldc=H * dk, a_kmajor=True, b_kmajor=True,
             batch=H, zdiv=1, a_s=(Mq * 152, 0), b_s=(dk, 0), c_s=(dk, 0), split_k=5, c_atomic=True)
    refp = torch.stack([A.double()[h, :, :P].t() @ Q.double()[:, h * dk:(h + 1) * dk] for h in range(H)], 1).reshape(P, H * dk)
    refp = refp.view(P, H, dk).reshape(P, H * dk)
    np.testing.assert_allclose(dp.cpu().double().numpy(), refp.numpy(), rtol=1e-3, atol=1e-3 * refp.abs().max().item())


@pytest.mark.parametrize("variant", ["tiles128", "tiles256_dma", "tiles256_dma_plain_loads", "tiles256_dma_all_nt_staggered"])
def test_grouped_wgrad_matches_reference(variant, monkeypatch):
    """csrc/gemm_grouped.hip through functional.flush_wgrads: several weight gradients of different shapes (ragged tiles,
    K tails, bias gradients, alpha, accumulation into non-zero dW) in one launch vs float64 — the register-staged
    128 x 128 kernel and the LDS-DMA fed 256 x 256 one (s2t_wgrad_grouped256; there also an FFN-sized problem, a K range
    that ends inside a 32-row step and operands that are column slices of a wider buffer)."""
    from s2t_amd import functional as Fn
    monkeypatch.setattr(Fn, "_WG_256", variant.startswith("tiles256_dma"))
    # the cache-policy / K-walk switches of the 256 x 256 kernel (read at every launch): same sums in another order at most
    if variant == "tiles256_dma_plain_loads":
        monkeypatch.setenv("S2T_WG_NT", "0")
    if variant == "tiles256_dma_all_nt_staggered":
        monkeypatch.setenv("S2T_WG_NT", "2")
        monkeypatch.setenv("S2T_WG_STAG", "3")
    g = torch.Generator().manual_seed(9)
    dev = "cuda"
    specs = [(300, 200, 2100, 1.0, True), (128, 128, 64, 0.5, False), (40, 520, 4000, 1.0, True), (256, 256, 6464, 2.0, True),
             (2048, 256, 9011, 1.0, True), (256, 2048, 4100, 0.5, True)]
    probs, refs = [], []
    for Nout, Kin, M, alpha, bias in specs:
        ldy, ldx = (Nout + 7) // 8 * 8, (Kin + 7) // 8 * 8
        dY = torch.zeros(M, ldy, dtype=torch.bfloat16); dY[:, :Nout] = _mk((M, Nout), torch.bfloat16, g)
        X = torch.zeros(M, ldx, dtype=torch.bfloat16); X[:, :Kin] = _mk((M, Kin), torch.bfloat16, g)
        dW = torch.full((Nout, Kin), 0.25, dtype=torch.float32, device=dev)
        db = torch.full((Nout,), -1.0, dtype=torch.float32, device=dev) if bias else None
        probs.append((dY.to(dev), X.to(dev), dW, Nout, Kin, M, ldy, ldx, alpha, db))
        refs.append((0.25 + alpha * dY[:, :Nout].double().t() @ X[:, :Kin].double(),
                     -1.0 + alpha * dY[:, :Nout].double().sum(0) if bias else None))
    # + a second problem accumulating into the first one's dW (tied weights): chained, one writer per element
    dY2 = _mk((1000, 304), torch.bfloat16, g); X2 = _mk((1000, 200), torch.bfloat16, g)
    dY2[:, 300:] = 0
    tied = (dY2.to(dev), X2.to(dev), probs[0][2], 300, 200, 1000, 304, 200, 3.0, None)
    refs[0] = (refs[0][0] + 3.0 * dY2[:, :300].double().t() @ X2.double(), refs[0][1])
    # + the three column slices of a fused [rows, 768] gradient against one input (the q / k / v projections)
    dqkv = _mk((3000, 768), torch.bfloat16, g).to(dev)
    xin = _mk((3000, 256), torch.bfloat16, g).to(dev)
    sl_w = [torch.zeros(256, 256, dtype=torch.float32, device=dev) for _ in range(3)]
    sl_b = [torch.zeros(256, dtype=torch.float32, device=dev) for _ in range(3)]
    slices = [(dqkv[:, 256 * i:], xin, sl_w[i], 256, 256, 3000, 768, 256, 1.0, sl_b[i]) for i in range(3)]
    Fn._WGQ["probs"] = list(probs) + [tied] + slices
    Fn.flush_wgrads()
    torch.cuda.synchronize()
    for i in range(3):
        rw = dqkv[:, 256 * i:256 * (i + 1)].double().t() @ xin.double()
        np.testing.assert_allclose(sl_w[i].cpu().double().numpy(), rw.cpu().numpy(), rtol=1e-3, atol=1e-3 * rw.abs().max().item())
        rb = dqkv[:, 256 * i:256 * (i + 1)].double().sum(0)
        np.testing.assert_allclose(sl_b[i].cpu().double().numpy(), rb.cpu().numpy(), rtol=1e-3, atol=1e-3 * rb.abs().max().item())
    for (dY, X, dW, Nout, Kin, M, ldy, ldx, alpha, db), (rw, rb) in zip(probs, refs):
        np.testing.assert_allclose(dW.cpu().double().numpy(), rw.numpy(), rtol=1e-3, atol=1e-3 * rw.abs().max().item())
        if db is not None:
            np.testing.assert_allclose(db.cpu().double().numpy(), rb.numpy(), rtol=1e-3, atol=1e-3 * rb.abs().max().item())


@pytest.mark.parametrize("M,N,K,split", [(1952, 256, 10000, 8), (300, 200, 2100, 4), (16000, 256, 4096, 2)])
def test_two_phase_splitk_with_bf16_result(M, N, K, split):
    """s2t_gemm, split_k > 1 with c_atomic = 2 and a bf16 C: fp32 partial tiles through the workspace, one rounding at the
    end (the input-gradient GEMMs with a long reduction over few output tiles: vocabulary projection, decoder FFN)."""
    g = torch.Generator().manual_seed(M + K)
    dev = "cuda"
    ldk, ldn = (K + 7) // 8 * 8, (N + 7) // 8 * 8
    A = torch.zeros(M, ldk, dtype=torch.bfloat16); A[:, :K] = _mk((M, K), torch.bfloat16, g)
    B = torch.zeros(K, ldn, dtype=torch.bfloat16); B[:, :N] = _mk((K, N), torch.bfloat16, g)
    out = torch.full((M, ldn), 9.0, dtype=torch.bfloat16, device=dev)
    ops.gemm(A.to(dev), B.to(dev), out, M=M, N=N, K=K, lda=ldk, ldb=ldn, ldc=ldn, b_kmajor=True, alpha=0.5, split_k=split,
             c_atomic=2)
    torch.cuda.synchronize()
    ref = 0.5 * A[:, :K].double() @ B[:, :N].double()
    got = out.cpu().double()[:, :N]
    assert (got - ref).abs().max() <= 1e-2 * ref.abs().max()
    if ldn > N:
        assert float((out.cpu().float()[:, N:] - 9.0).abs().max()) == 0.0  # columns beyond N untouched


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K,split,act", [(3904, 256, 2048, 8, None), (300, 200, 1100, 3, "relu"), (130, 136, 640, 5, "swish")])
def test_two_phase_splitk_runs_the_fused_epilogue(dtype, M, N, K, split, act):
    """c_atomic = 2 with bias / activation (+ pre-activation output) / dropout / row mask / alpha / residual: the second
    phase runs the epilogue of the one-pass kernel on the summed tiles — same dropout mask, values equal up to the order
    of the fp32 sums (one bf16 ulp after rounding)."""
    g = torch.Generator().manual_seed(M + K)
    dev = "cuda"
    ld = (lambda n: _pad_ld(n, dtype))
    A = torch.zeros(M, ld(K), dtype=dtype); A[:, :K] = _mk((M, K), dtype, g, 0.5)
    B = torch.zeros(N, ld(K), dtype=dtype); B[:, :K] = _mk((N, K), dtype, g, 0.1)
    bias = _mk((N,), torch.float32, g)
    res = torch.zeros(M, ld(N), dtype=dtype); res[:, :N] = _mk((M, N), dtype, g)
    T = 50
    nb = (M + T - 1) // T
    lens = torch.randint(T // 2, T + 1, (nb,), generator=g).to(torch.int32).to(dev)
    seed = torch.tensor([1234], dtype=torch.int64, device=dev)
    outs = []
    for sk in (1, split):
        out = torch.full((M, ld(N)), 9.0, dtype=dtype, device=dev)
        pre = torch.full((M, ld(N)), 9.0, dtype=dtype, device=dev) if act else None
        ops.gemm(A.to(dev), B.to(dev), out, M=M, N=N, K=K, lda=ld(K), ldb=ld(K), ldc=ld(N), bias=bias.to(dev), act=act,
                 alpha=0.5, residual=res.to(dev), ldr=ld(N), preact=pre, ldp=ld(N), row_lens=lens, row_T=T,
                 drop=(0.1, seed, 7), split_k=sk, c_atomic=2 if sk > 1 else False)
        torch.cuda.synchronize()
        outs.append((out.cpu().double(), pre.cpu().double() if act else None))
    (o1, p1), (o2, p2) = outs
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert (o1[:, :N] - o2[:, :N]).abs().max() <= tol * o1[:, :N].abs().max()
    # the dropped / masked positions (value == residual exactly) coincide
    r = res.double()[:, :N]
    assert ((o1[:, :N] == r) == (o2[:, :N] == r)).float().mean() > 0.999
    assert ((o1[:, :N] == r).float().mean() > 0.05)
    if act:
        assert (p1[:, :N] - p2[:, :N]).abs().max() <= tol * p1[:, :N].abs().max()
    if ld(N) > N:
        assert float((o2[:, N:] - 9.0).abs().max()) == 0.0
    # against the float64 product
    v = A[:, :K].double() @ B[:, :K].double().t() + bias.double()
    if act == "relu":
        v = v.clamp(min=0)
    elif act == "swish":
        v = v * torch.sigmoid(v)
    rows = torch.arange(M)
    masked = (rows % T) >= lens.cpu()[rows // T]
    kept = (o2[:, :N] != r)
    ref = 0.5 * v / 0.9 + r
    err = ((o2[:, :N] - ref).abs() * kept)[~masked]
    assert err.max() <= (3e-2 if dtype == torch.bfloat16 else 1e-4) * ref.abs().max()
    assert float((o2[:, :N] - r)[masked].abs().max()) == 0.0


def test_overwrite_splitk_without_a_workspace_runs_as_one_pass():
    """c_atomic = 2 with no workspace handed over: s2t_gemm runs the one-pass kernel (split_k = 1) — bit-identical to it."""
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    M, N, K = 260, 136, 1024
    A = _mk((M, K), torch.bfloat16, g, 0.5).to(dev)
    B = _mk((N, K), torch.bfloat16, g, 0.1).to(dev)
    bias = _mk((N,), torch.float32, g).to(dev)
    res = _mk((M, N), torch.bfloat16, g).to(dev)
    outs = []
    for kw in (dict(), dict(split_k=4, c_atomic=2, splitk_workspace=False)):
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(A, B, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, act="relu", alpha=0.7, residual=res, ldr=N, **kw)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])


# ---- the 256 x 256 LDS-DMA path (gemm256.hip): bf16, row-major A and B, K % 64 == 0 ----------------------------------------
class _large_tile:
    """Pin s2t_gemm's large-tile switch for a block (s2t_gemm_configure): 0 never, 2 / 3 whenever the arguments allow with
    256- / 128-row tiles."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = ops.gemm_configure()
        ops.gemm_configure(self.mode)

    def __exit__(self, *a):
        ops.gemm_configure(self.prev)


def _both_paths(call):
    """(128 x 128 path, 256-row large tiles); the 128-row large tiles (mode 3) are checked against them on the way."""
    outs = []
    for mode in (0, 2, 3):
        with _large_tile(mode):
            outs.append(call())
    torch.cuda.synchronize()
    assert torch.equal(outs[1], outs[2]), "128-row and 256-row large tiles differ"
    return outs[:2]


@pytest.mark.parametrize("M,N,K,ldpad", [(256, 256, 128, 0), (1000, 520, 192, 0), (513, 264, 64 * 5, 8), (4100, 777 * 8, 256, 0),
                                         (16000, 768, 256, 0), (257, 10000, 512, 0), (300, 264, 200, 8), (700, 512, 10000, 0),
                                         (515, 1000, 136, 0)])
@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("bkm", [False, True])
def test_large_tile_plain(M, N, K, ldpad, cdt, bkm):
    """Ragged M, N and K (clamped duplicate rows / columns are never stored; the K tail is fetched from beyond the buffer
    descriptors' ranges and reads as zero — the padding behind it is poisoned here), padded leading dimensions, both layouts of
    B, both output types: equal to the 128 x 128 path bit for bit (same MFMA, same order over K) and to a float64 product
    within bf16 rounding."""
    g = torch.Generator().manual_seed(M + N + K)
    dev = "cuda"
    lda, ldc = K + ldpad, N + ldpad
    A = torch.full((M, lda), float("nan"), dtype=torch.bfloat16); A[:, :K] = _mk((M, K), torch.bfloat16, g)
    Wl = _mk((N, K), torch.bfloat16, g, K ** -0.5)
    if bkm:
        ldb = N + 2 * ldpad
        W = torch.full((K, ldb), float("nan"), dtype=torch.bfloat16); W[:, :N] = Wl.t()
    else:
        ldb = K + 2 * ldpad
        W = torch.full((N, ldb), float("nan"), dtype=torch.bfloat16); W[:, :K] = Wl
    Ad, Wd = A.to(dev), W.to(dev)

    def call():
        out = torch.full((M, ldc), 7.0, dtype=cdt, device=dev)
        ops.gemm(Ad, Wd, out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, b_kmajor=bkm)
        return out

    with _large_tile(2):
        a = _gemm_args(Ad, Wd, call(), M, N, K, lda, ldb, ldc)
        a.b_kmajor = int(bkm)
        assert "gemm256_kernel" in ops.gemm_symbol(a)
    old, new = _both_paths(call)
    assert torch.equal(old, new)
    if ldpad:
        assert (new[:, N:] == 7.0).all(), "wrote outside the N columns"
    if M * N <= 4100 * 6216:
        ref = A[:, :K].double() @ Wl.double().t()
        np.testing.assert_allclose(new[:, :N].cpu().double().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2 * ref.abs().max().item())


def _gemm_args(A, B, out, M, N, K, lda, ldb, ldc):
    from s2t_amd import _lib as L
    a = L.GemmArgs()
    a.dtype, a.c_dtype = L.dtype_id(A.dtype), L.dtype_id(out.dtype)
    a.M, a.N, a.K = M, N, K
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc
    a.batch = a.zdiv = a.split_k = 1
    return a


@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
def test_large_tile_epilogue(cdt):
    """Every epilogue stage on the large tile against the 128 x 128 path (bit for bit: both follow Epi::finish's order) and a
    float64 restatement: bias, pre-activation copy, ReLU / swish, act'(z), dropout (the same counter-based mask), alpha, the
    padded-frame mask in both row geometries, residual."""
    g = torch.Generator().manual_seed(11)
    dev = "cuda"
    B_, T, K, N = 5, 150, 256, 520
    M = B_ * T
    A = _mk((M, K), torch.bfloat16, g).to(dev); W = _mk((N, K), torch.bfloat16, g, K ** -0.5).to(dev)
    bias = _mk((N,), torch.float32, g).to(dev)
    R = _mk((M, N), cdt, g).to(dev)
    Z = _mk((M, N), cdt, g).to(dev)
    lens = torch.tensor([150, 31, 7, 149, 1], dtype=torch.int32, device=dev)
    seed = torch.tensor([1234], dtype=torch.int64, device=dev)
    mask = (torch.arange(T, device=dev)[None, :] >= lens[:, None]).reshape(-1)
    for act in ("relu", "swish", None):
        def call():
            out = torch.empty(M, N, dtype=cdt, device=dev)
            pre = torch.zeros(M, N, dtype=cdt, device=dev)
            ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, act=act, alpha=0.5, residual=R, ldr=N,
                     preact=pre if act else None, ldp=N, row_lens=lens, row_T=T)
            return torch.cat([out, pre])
        old, new = _both_paths(call)
        assert torch.equal(old, new), act
        z = A.double() @ W.double().t() + bias.double()
        a = {"relu": torch.relu, "swish": lambda v: v * torch.sigmoid(v), None: lambda v: v}[act](z)
        br = 0.5 * a
        br[mask] = 0
        ref = R.double() + br
        np.testing.assert_allclose(new[:M].double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=4e-2)
        if act:
            np.testing.assert_allclose(new[M:].double().cpu().numpy(), z.cpu().numpy(), rtol=1e-2, atol=4e-2)
    for dact in ("relu", "swish"):
        def call():
            out = torch.empty(M, N, dtype=cdt, device=dev)
            ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, dact_z=Z, ldz=N, dact=dact, drop=(0.25, seed, 3))
            return out
        old, new = _both_paths(call)
        assert torch.equal(old, new), dact
        zd = Z.double()
        d = (zd > 0).double() if dact == "relu" else torch.sigmoid(zd) * (1 + zd * (1 - torch.sigmoid(zd)))
        keep = (new[d != 0] != 0).double().mean().item()
        assert 0.72 < keep < 0.78
        ref = (A.double() @ W.double().t()) * d / 0.75
        got = new.double()
        kept = got != 0
        np.testing.assert_allclose(got[kept].cpu().numpy(), ref[kept].cpu().numpy(), rtol=1e-2, atol=4e-2)


@pytest.mark.parametrize("M,N,K,split,bkm", [(16000, 256, 10000, 2, True), (3904, 256, 10000, 8, True), (300, 264, 4104, 4, True),
                                             (1000, 520, 2048, 3, False), (257, 136, 10000, 7, False), (3904, 256, 2048, 4, True)])
@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
def test_large_tile_split_k(M, N, K, split, bkm, cdt):
    """Two-phase split-K (c_atomic = 2, workspace) on the large tiles: the (tile, split) pairs are the work items, the partial
    tiles land in the 128 x 128 kernel's workspace order and ITS second phase sums them — ragged M / N / K, a last split shorter
    than the others (its steps beyond K fetch zeros), the fused second-phase epilogue (bias + residual, alpha), both output types:
    equal to the 128 x 128 path bit for bit."""
    import ctypes
    from s2t_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K + split)
    dev = "cuda"
    A = _mk((M, K), torch.bfloat16, g)
    Wl = _mk((N, K), torch.bfloat16, g, K ** -0.5)
    Ad = A.to(dev)
    Wd = (Wl.t().contiguous() if bkm else Wl).to(dev)
    ldb = N if bkm else K
    bias = _mk((N,), torch.float32, g).to(dev)
    res = _mk((M, N), cdt, g).to(dev)
    for epi in (False, True):
        def call():
            out = torch.full((M, N), 7.0, dtype=cdt, device=dev)
            kw = dict(bias=bias, residual=res, ldr=N) if epi else {}
            ops.gemm(Ad, Wd, out, M=M, N=N, K=K, lda=K, ldb=ldb, ldc=N, b_kmajor=bkm, alpha=0.5, split_k=split, c_atomic=2, **kw)
            return out

        old, new = _both_paths(call)
        assert torch.equal(old, new)
        ref = 0.5 * (A.double() @ Wl.double().t() + (bias.cpu().double() if epi else 0.0)) + (res.cpu().double() if epi else 0.0)
        np.testing.assert_allclose(new.cpu().double().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2 * ref.abs().max().item())
    # the large-tile kernel is the one that ran (at least eight K-steps per split)
    with _large_tile(2):
        out = torch.empty(M, N, dtype=cdt, device=dev)
        a = _gemm_args(Ad, Wd, out, M, N, K, K, ldb, N)
        a.b_kmajor, a.split_k, a.c_atomic = int(bkm), split, 2
        need = L.lib().s2t_gemm_ws_floats(ctypes.byref(a))
        ws = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
        a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
        sym = ops.gemm_symbol(a)
        assert ("gemm256_kernel" in sym and sym.endswith("true>")) == (K >= 512 * split), sym


def test_large_tile_packed_rows():
    """row_T = S2T_ROWS_PACKED: the live row count is read on the device (row blocks beyond it are never walked, rows beyond it
    never stored) and halo rows come out zero — as on the 128 x 128 path."""
    from s2t_amd import rows as Rows
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    B_, T, K, N = 40, 120, 256, 768
    lens = torch.randint(20, T + 1, (B_,), generator=g).to(torch.int32).to(dev)
    Rows.attach(lens, B_, T, 7, tag="test_large_tile")
    M = B_ * T
    live = lens._pk.live_rows()
    assert live < M - 300
    A = _mk((M, K), torch.bfloat16, g).to(dev); W = _mk((N, K), torch.bfloat16, g, K ** -0.5).to(dev)

    def call():
        out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
        ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, row_lens=lens, row_T=T)
        return out
    old, new = _both_paths(call)
    assert torch.equal(old, new)
    assert (new[live:] == 7.0).all()
    m = lens._pk.row_map[:live]
    assert (new[:live][m < 0] == 0).all()
    ref = (A[:live].float() @ W.float().t())
    np.testing.assert_allclose(new[:live][m >= 0].float().cpu().numpy(), ref[m >= 0].cpu().numpy(), rtol=1e-2, atol=4e-2)


def test_large_tile_split_k_packed_rows():
    """Two-phase split-K over a packed batch (the CTC head's input gradient): the work items cover the LIVE row blocks only, the
    second phase reads exactly those partial tiles, rows beyond the live count keep what they held — bit for bit the 128 x 128
    path, with both tile heights."""
    from s2t_amd import rows as Rows
    dev = "cuda"
    g = torch.Generator().manual_seed(5)
    B_, T, K, N = 24, 130, 5000 + 8, 256
    lens = torch.randint(30, T + 1, (B_,), generator=g).to(torch.int32).to(dev)
    Rows.attach(lens, B_, T, 7, tag="test_large_tile_splitk")
    M = B_ * T
    live = lens._pk.live_rows()
    assert live < M - 200
    A = _mk((M, K), torch.bfloat16, g).to(dev); W = _mk((K, N), torch.bfloat16, g, K ** -0.5).to(dev)

    def call():
        out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
        ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, b_kmajor=True, split_k=4, c_atomic=2, rows=lens)
        return out
    old, new = _both_paths(call)
    assert torch.equal(old, new)
    assert (new[live:] == 7.0).all()
    ref = A[:live].float() @ W.float()
    np.testing.assert_allclose(new[:live].float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=4e-2)


@pytest.mark.parametrize("cdt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,nout,K,stride", [(2000, 512, 400, 160), (700, 136, 192, 0), (4100, 256, 2560, 1024)])
def test_large_tile_glu(cdt, M, nout, K, stride):
    """The GLU form (value | gate weight rows, pre-activation halves saved) on the large tile, A rows overlapping as in the
    subsampler's convolutions (row m starts stride elements behind row m - 1): bit for bit the 128 x 128 path's result, and
    (a + b_a) * sigmoid(g + b_g) in float64."""
    g = torch.Generator().manual_seed(M + nout)
    dev = "cuda"
    lda = stride if stride else K
    flat = _mk(((M - 1) * lda + K,), torch.bfloat16, g).to(dev)
    W = _mk((2 * nout, K), torch.bfloat16, g, K ** -0.5).to(dev)
    bias = _mk((2 * nout,), torch.float32, g).to(dev)
    R = _mk((M, nout), cdt, g).to(dev)

    def call():
        out = torch.empty(M, nout, dtype=cdt, device=dev)
        pre = torch.zeros(M, 2 * nout, dtype=cdt, device=dev)
        ops.gemm(flat, W, out, M=M, N=2 * nout, K=K, lda=lda, ldb=K, ldc=nout, bias=bias, act="glu", preact=pre, ldp=2 * nout,
                 residual=R, ldr=nout, alpha=0.5)
        return torch.cat([out, pre], 1)
    old, new = _both_paths(call)
    assert torch.equal(old, new)
    A = torch.as_strided(flat, (M, K), (lda, 1)).double()
    z = A @ W.double().t() + bias.double()
    ref = R.double() + 0.5 * z[:, :nout] * torch.sigmoid(z[:, nout:])
    np.testing.assert_allclose(new[:, :nout].double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=4e-2)
    np.testing.assert_allclose(new[:, nout:].double().cpu().numpy(), z.cpu().numpy(), rtol=1e-2, atol=4e-2)


def test_large_tile_batched_conv_rows():
    """A batch folded into the row-block walk (the subsampler's per-utterance convolutions: overlapping A rows, GLU, padded-frame
    mask per utterance, dropout indices over the global row): equal to the 128 x 128 path bit for bit."""
    g = torch.Generator().manual_seed(9)
    dev = "cuda"
    Bz, Tp, Cin, Kw, stride, Cout = 40, 1000, 80, 5, 2, 512
    Tout = (Tp - Kw) // stride + 1
    x = _mk((Bz, Tp * Cin), torch.bfloat16, g).to(dev)
    W = _mk((Cout, Kw * Cin), torch.bfloat16, g, (Kw * Cin) ** -0.5).to(dev)
    b = _mk((Cout,), torch.float32, g).to(dev)
    lens = torch.randint(100, Tout + 1, (Bz,), generator=g).to(torch.int32).to(dev)
    seed = torch.tensor([77], dtype=torch.int64, device=dev)
    for act, nout in (("glu", Cout // 2), ("relu", Cout)):
        def call():
            out = torch.full((Bz, Tout, nout), 7.0, dtype=torch.bfloat16, device=dev)
            ops.gemm(x, W, out, M=Tout, N=Cout, K=Kw * Cin, lda=stride * Cin, ldb=Kw * Cin, ldc=nout, batch=Bz, a_s=(Tp * Cin, 0),
                     c_s=(Tout * nout, 0), bias=b, act=act, row_lens=lens, row_T=Tout, drop=(0.1, seed, 5) if act == "relu" else None)
            return out
        old, new = _both_paths(call)
        assert torch.equal(old, new), act
        t = torch.arange(Tout, device=dev)[None] >= lens[:, None]
        assert (new[t] == 0).all()
        A = torch.as_strided(x, (Bz, Tout, Kw * Cin), (Tp * Cin, stride * Cin, 1)).double()
        z = A @ W.double().t() + b.double()
        ref = z[..., :nout] * torch.sigmoid(z[..., nout:]) if act == "glu" else torch.relu(z) / 0.9
        got = new.double()
        keep = (~t)[..., None] & (got != 0)
        np.testing.assert_allclose(got[keep].cpu().numpy(), ref[keep].cpu().numpy(), rtol=1e-2, atol=4e-2)


def test_large_tile_random_shapes_equal_the_small_tile_path():
    """Forty random problems — M, N, K (multiples of 8, K >= 128), leading-dimension padding, B layout, output type, bias /
    activation / residual / alpha / mask / dropout drawn at random — through both tile paths: equal bit for bit."""
    rng = np.random.RandomState(7)
    dev = "cuda"
    seed = torch.tensor([99], dtype=torch.int64, device=dev)
    for case in range(40):
        M = int(rng.randint(1, 1400))
        N = 8 * int(rng.randint(1, 140))
        K = 8 * int(rng.randint(16, 90))
        bkm = bool(rng.randint(2))
        cdt = torch.float32 if rng.randint(3) == 0 else torch.bfloat16
        pad = 8 * int(rng.randint(0, 3))
        g = torch.Generator().manual_seed(1000 + case)
        lda, ldc = K + pad, N + pad
        A = torch.zeros(M, lda, dtype=torch.bfloat16); A[:, :K] = _mk((M, K), torch.bfloat16, g)
        Wl = _mk((N, K), torch.bfloat16, g, K ** -0.5)
        if bkm:
            ldb = N + pad
            W = torch.zeros(K, ldb, dtype=torch.bfloat16); W[:, :N] = Wl.t()
        else:
            ldb = K + pad
            W = torch.zeros(N, ldb, dtype=torch.bfloat16); W[:, :K] = Wl
        Ad, Wd = A.to(dev), W.to(dev)
        kw = {}
        if rng.randint(2):
            kw["bias"] = _mk((N,), torch.float32, g).to(dev)
        act = [None, "relu", "swish"][rng.randint(3)]
        if act:
            kw["act"] = act
        if rng.randint(2):
            kw["residual"] = _mk((M, ldc), cdt, g).to(dev); kw["ldr"] = ldc
        if rng.randint(2):
            kw["alpha"] = 0.5
        if rng.randint(3) == 0:
            kw["drop"] = (0.2, seed, case)
        if rng.randint(3) == 0:
            T = int(rng.randint(1, 60))
            Bz = (M + T - 1) // T
            kw["row_lens"] = torch.randint(0, T + 1, (Bz,), generator=g).to(torch.int32).to(dev); kw["row_T"] = T

        def call():
            out = torch.full((M, ldc), 7.0, dtype=cdt, device=dev)
            ops.gemm(Ad, Wd, out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, b_kmajor=bkm, **kw)
            return out
        old, new = _both_paths(call)
        assert torch.equal(old, new), (case, M, N, K, bkm, str(cdt), sorted(kw))
