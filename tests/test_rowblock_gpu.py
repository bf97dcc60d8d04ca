"""GPU parity of the row-block kernels (csrc/rowblock.hip) through the C-ABI.

s2t_ffn_fused_fwd is compared with (a) plain fp32 maths on the same bf16-rounded operands — the reference's
FeedForwardModule (fairseq/modules/s2t_transformer_layer.py:55-66) between its LayerNorm and residual — and (b) the
unfused kernels (s2t_layernorm_fwd + two s2t_gemm) for the dropout masks, which must be the SAME masks element for
element because the unfused backward regenerates them."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from s2t_amd import kernels as K  # noqa: E402

DEV = "cuda"


@pytest.fixture
def ffn_split():
    """Pin / cap the workgroups per 128-row block of the fused feed-forward kernels for one test (s2t_ffn_configure)."""
    _, old, _ = K.ffn_configure()

    def set_(n):
        K.ffn_configure(split=n)

    yield set_
    K.ffn_configure(split=old)


def _ffn_name(mode_bwd=False):
    """Kernel name of the LAST profiled fused feed-forward launch (what rocprofv3 prints)."""
    return K.GEMM_PROFILE[-1][0]


def _mk(M, F, seed, d=256):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, d, generator=g).bfloat16()
    w1 = (torch.randn(F, d, generator=g) * d ** -0.5).bfloat16()
    b1 = 0.1 * torch.randn(F, generator=g)
    w2 = (torch.randn(d, F, generator=g) * F ** -0.5).bfloat16()
    b2 = 0.1 * torch.randn(d, generator=g)
    gam = 1 + 0.1 * torch.randn(d, generator=g)
    bet = 0.1 * torch.randn(d, generator=g)
    eg = 1 + 0.1 * torch.randn(d, generator=g)
    eb = 0.1 * torch.randn(d, generator=g)
    return x, w1, b1, w2, b2, gam, bet, eg, eb


def _act(v, act):
    if act == "relu":
        return torch.relu(v)
    if act == "swish":
        return v * torch.sigmoid(v)
    return v


def _ref(x, w1, b1, w2, b2, gam, bet, eg, eb, act, alpha, use_ln, use_eln, lens=None, T=0):
    """fp32 maths with the rounding points of the bf16 contract: LN output, hidden activation and block output are bf16."""
    xf = x.float()
    xn = torch.nn.functional.layer_norm(xf, (xf.shape[1],), gam, bet, 1e-5).bfloat16().float() if use_ln else xf
    z = xn @ w1.float().t() + b1
    h = _act(z, act).bfloat16().float()
    y = (xf + alpha * (h @ w2.float().t() + b2)).bfloat16().float()
    yl = None
    if use_eln:
        yl = torch.nn.functional.layer_norm(y, (y.shape[1],), eg, eb, 1e-5)
        if lens is not None:
            B = lens.numel()
            pad = (torch.arange(T)[None] >= lens[:, None]).reshape(-1)
            yl = yl.masked_fill(pad[:, None], 0.0)
    return xn, z, h, y, yl


@pytest.mark.parametrize("M,F,act,use_ln,use_eln", [
    (64, 64, "relu", False, False),
    (64, 128, "swish", True, False),
    (200, 2048, "swish", True, True),     # ragged last row block
    (1000, 512, "relu", True, True),
    (16000, 2048, "swish", True, True),   # the headline configuration's encoder FFN
])
def test_ffn_fused_eval(M, F, act, use_ln, use_eln):
    x, w1, b1, w2, b2, gam, bet, eg, eb = _mk(M, F, 7 + F)
    T = 50
    lens = None
    if use_eln and M % T == 0:
        B = M // T
        lens = torch.randint(1, T + 1, (B,), generator=torch.Generator().manual_seed(3)).int()
    alpha = 0.5
    _, _, _, y_ref, yl_ref = _ref(x, w1, b1, w2, b2, gam, bet, eg, eb, act, alpha, use_ln, use_eln, lens, T)
    xd = x.to(DEV)
    y = torch.full((M, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
    yl = torch.full((M, 256), float("nan"), dtype=torch.bfloat16, device=DEV) if use_eln else None
    K.ffn_fused_fwd(xd, w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), y, act=act, alpha=alpha, residual=xd,
                    ln=(gam.to(DEV), bet.to(DEV)) if use_ln else None,
                    end_ln=(eg.to(DEV), eb.to(DEV)) if use_eln else None, y_ln=yl,
                    end_lens=lens.to(DEV) if lens is not None else None, end_T=T)
    torch.cuda.synchronize()
    err = (y.float().cpu() - y_ref).abs().max() / y_ref.abs().max()
    assert float(err) < 1.5e-2, float(err)
    if use_eln:
        err = (yl.float().cpu() - yl_ref).abs().max() / yl_ref.abs().max()
        assert float(err) < 2.5e-2, float(err)
        if lens is not None:
            pad = (torch.arange(T)[None] >= lens[:, None]).reshape(-1)
            assert float(yl.float().cpu()[pad].abs().max()) == 0.0


def _untile_z(zt, M, F):
    """Row-major [M, F] view of a z buffer in the tiled layout of include/s2t_hip.h (s2t_ffn_args.z_tiled_ok)."""
    P = (M + 127) // 128
    t = zt.reshape(-1)[: P * 128 * F].view(P, F // 64, 4, 4, 2, 32, 8)   # [p][cg][wi][s][hh][m][j]
    return t.permute(0, 2, 5, 1, 3, 4, 6).reshape(P * 128, F)[:M]        # row = 128 p + 32 wi + m, unit = 64 cg + 16 s + 8 hh + j


@pytest.mark.parametrize("M,F,act,tiled,split", [
    (200, 256, "swish", False, None), (4096, 2048, "relu", False, None), (200, 256, "swish", True, None),
    (4096, 2048, "swish", True, None), (1000, 512, "relu", True, None),
    # the instantiations the headline bench runs: ffn_pc_kernel<1, 2, true, 2> (two workgroups per block: 65-128 blocks, or
    # capped), the four-part form at 8 000 rows, and one workgroup per block
    (16000, 2048, "swish", True, None), (4096, 2048, "swish", True, 2), (8000, 2048, "swish", True, None),
    (4096, 2048, "swish", True, 1)])
def test_ffn_fused_train_saves_and_masks(M, F, act, tiled, split, ffn_split):
    """Training flavour: the saves equal what the unfused kernels produce on the same inputs with the same dropout sites.
    ``tiled``: the caller accepts z in the 128-row kernel's tiled layout (that kernel then runs the training forward).
    ``split``: workgroups per 128-row block pinned / capped (None: chosen from the row count)."""
    if split is not None:
        ffn_split(split)
    x, w1, b1, w2, b2, gam, bet, eg, eb = _mk(M, F, 11)
    d = 256
    alpha = 0.5
    xd, w1d, b1d, w2d, b2d = x.to(DEV), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV)
    gd, bd = gam.to(DEV), bet.to(DEV)
    seed = torch.tensor([12345], dtype=torch.int64, device=DEV)
    drop_h, drop_o = (0.1, seed, 3), (0.1, seed, 4)
    # unfused: LayerNorm kernel, GEMM (bias + act + preact + dropout), GEMM (bias + dropout + alpha + residual)
    xl_u = torch.empty_like(xd)
    mean_u, rstd_u = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    K.layernorm_fwd(xd, gd, bd, xl_u, mean_u, rstd_u, M, d)
    h_u = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    z_u = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    K.gemm(xl_u, w1d, h_u, M=M, N=F, K=d, lda=d, ldb=d, ldc=F, bias=b1d, act=act, preact=z_u, ldp=F, drop=drop_h)
    y_u = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    K.gemm(h_u, w2d, y_u, M=M, N=d, K=F, lda=F, ldb=F, ldc=d, bias=b2d, alpha=alpha, residual=xd, ldr=d, drop=drop_o)
    # fused
    xl_f = torch.empty_like(xd)
    mean_f, rstd_f = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    h_f = torch.full((M, F), float("nan"), dtype=torch.bfloat16, device=DEV)
    z_f = torch.full((K.ffn_z_rows(M) if tiled else M, F), float("nan"), dtype=torch.bfloat16, device=DEV)
    y_f = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    K.GEMM_PROFILE = []
    try:
        was_tiled = K.ffn_fused_fwd(xd, w1d, b1d, w2d, b2d, y_f, act=act, alpha=alpha, residual=xd, ln=(gd, bd), x_ln=xl_f,
                                    ln_stats=(mean_f, rstd_f), z=z_f, h=h_f, drop_h=drop_h, drop_o=drop_o, z_tiled_ok=tiled)
        torch.cuda.synchronize()
        name = _ffn_name()
    finally:
        K.GEMM_PROFILE = None
    K.ffn_exchange_check()
    assert was_tiled == (tiled and bool(K.ffn_configure()[0] & 2))
    if was_tiled and M in (16000, 8000, 4096) and F == 2048:  # the launch really was the form this case is about
        want = {(16000, None): 2, (8000, None): 4, (4096, 2): 2, (4096, 1): 1, (4096, None): 8}[(M, split)]
        assert name == "ffn_pc_kernel<1, %d, true, %d>" % (2 if act == "swish" else 1, want), name
    if was_tiled:
        z_f = _untile_z(z_f, M, F)
    assert float((mean_f - mean_u).abs().max()) < 1e-5
    assert float(((rstd_f - rstd_u) / rstd_u).abs().max()) < 1e-5
    assert float((xl_f.float() - xl_u.float()).abs().max()) <= 2 ** -6  # one bf16 ulp at |x| < 4
    zf, zu = z_f.float(), z_u.float()
    assert float((zf - zu).abs().max() / zu.abs().max()) < 2e-2
    # identical masks: an element is zero in one exactly when it is zero in the other (apart from true zeros of relu)
    hf, hu = h_f.float(), h_u.float()
    nz_f, nz_u = hf != 0, hu != 0
    mism = (nz_f != nz_u) & (zu.abs() > 0.05)
    assert int(mism.sum()) == 0, int(mism.sum())
    keep = float(nz_u.float().mean())
    assert keep < (0.96 if act == "swish" else 0.6)  # dropout really was applied
    assert float((hf - hu).abs().max() / hu.abs().max()) < 2e-2
    assert float((y_f.float() - y_u.float()).abs().max() / y_u.float().abs().max()) < 2e-2


def _tr_table(pairs):
    import numpy as np

    rec = np.zeros(len(pairs), dtype=np.dtype([("src", "u8"), ("dst", "u8"), ("rows", "i4"), ("cols", "i4")]))
    for i, (s_, d_) in enumerate(pairs):
        rec[i] = (s_.data_ptr(), d_.data_ptr(), s_.shape[0], s_.shape[1])
    return torch.from_numpy(rec.view(np.uint8).copy()).to(DEV)


def test_transpose_batched():
    """s2t_transpose_bf16_batched: several matrices (one ragged, one with odd extents) by one launch, bit exact."""
    g = torch.Generator().manual_seed(2)
    shapes = [(2048, 256), (256, 2048), (100, 72), (65, 33)]
    src = [torch.randn(r, c, generator=g).bfloat16().to(DEV) for r, c in shapes]
    dst = [torch.full((c, r), 7.0, dtype=torch.bfloat16, device=DEV) for r, c in shapes]
    tl = K.transpose_tiles(shapes)
    K.transpose_batched(_tr_table(list(zip(src, dst))), len(src), torch.from_numpy(tl).to(DEV), int(tl.shape[0]))
    torch.cuda.synchronize()
    for s_, d_ in zip(src, dst):
        assert torch.equal(d_, s_.t().contiguous())


def _tile_z(z, M, F):
    """The inverse of _untile_z: a row-major [M, F] z into the tiled layout (rows padded to a multiple of 128)."""
    P = (M + 127) // 128
    full = torch.zeros(P * 128, F, dtype=z.dtype, device=z.device)
    full[:M] = z
    return full.view(P, 4, 32, F // 64, 4, 2, 8).permute(0, 3, 1, 4, 5, 2, 6).contiguous().view(P * 128, F)


@pytest.mark.parametrize("M,F,act,p,tiled", [(64, 64, "relu", 0.0, False), (200, 256, "swish", 0.1, False), (4096, 2048, "swish", 0.1, False),
                                             (1000, 512, "relu", 0.15, False), (130, 128, "none", 0.0, False),
                                             (200, 256, "swish", 0.1, True), (4096, 2048, "swish", 0.1, True), (1000, 512, "relu", 0.15, True),
                                             # ffn_pc_kernel<2, 2, true, 2> (the bench's backward), the four-part form, one workgroup per block
                                             (16000, 2048, "swish", 0.1, True), (8000, 2048, "swish", 0.1, True),
                                             (4096, 2048, "swish", 0.1, (True, 2)), (4096, 2048, "swish", 0.1, (True, 1))])
def test_ffn_fused_bwd_matches_the_two_dgrad_gemms(M, F, act, p, tiled, ffn_split):
    """s2t_ffn_fused_bwd against (a) fp32 maths on the same bf16 operands and (b) the unfused s2t_gemm pair it replaces:
    dz = alpha * drop_h((dy W2) * act'(z)), dxn = dz W1 — the autograd backward of the two F.linear, the activation and the
    hidden dropout of modules/s2t_transformer_layer.py:55-66.  (b) shares the dropout mask, so dz agrees to bf16 rounding."""
    from s2t_amd import functional as Fn

    split = None
    if isinstance(tiled, tuple):
        tiled, split = tiled
        ffn_split(split)
    d = 256
    g = torch.Generator().manual_seed(M + F)
    dy = torch.randn(M, d, generator=g).bfloat16().to(DEV)
    z = torch.randn(M, F, generator=g).bfloat16().to(DEV)
    w1 = (torch.randn(F, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
    w2 = (torch.randn(d, F, generator=g) * F ** -0.5).bfloat16().to(DEV)
    alpha = 0.5
    w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
    Fn.DROPOUT.begin_step(torch.device(DEV))
    Fn.DROPOUT.set_seed(31)
    drop = Fn.DROPOUT.next(p, torch.device(DEV))
    dz = torch.full((M, F), 3.0, dtype=torch.bfloat16, device=DEV)
    dxn = torch.full((M, d), 3.0, dtype=torch.bfloat16, device=DEV)
    K.GEMM_PROFILE = []
    try:
        K.ffn_fused_bwd(dy, w2t, w1t, _tile_z(z, M, F) if tiled else z, dz, dxn, act=act, alpha=alpha, drop_h=drop, z_tiled=tiled)
        torch.cuda.synchronize()
        name = _ffn_name()
    finally:
        K.GEMM_PROFILE = None
    K.ffn_exchange_check()
    if tiled and F == 2048 and M in (16000, 8000, 4096) and (K.ffn_configure()[0] & 4):
        want = {(16000, None): 2, (8000, None): 4, (4096, 2): 2, (4096, 1): 1, (4096, None): 8}[(M, split)]
        assert name == "ffn_pc_kernel<2, 2, true, %d>" % want, name
    # (b) the unfused pair
    dz_u = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    dx_u = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    K.gemm(dy, w2, dz_u, M=M, N=F, K=d, lda=d, ldb=F, ldc=F, b_kmajor=True, alpha=alpha, dact_z=z, ldz=F, dact=act, drop=drop)
    K.gemm(dz_u, w1, dx_u, M=M, N=d, K=F, lda=F, ldb=d, ldc=d, b_kmajor=True)
    torch.cuda.synchronize()
    assert ((dz.float() == 0) == (dz_u.float() == 0)).float().mean() > 0.999  # same mask (exact zeros of relu' aside)
    err = (dz.float() - dz_u.float()).abs().max() / dz_u.float().abs().max()
    assert err < 1e-2, float(err)
    rel = (dxn.float() - dx_u.float()).norm() / dx_u.float().norm()
    assert rel < 1e-2, float(rel)
    # (a) fp32 maths, with the kernel's own mask (the zeros of dz_u where act' is not zero)
    zf = z.float()
    if act == "relu":
        da = (zf > 0).float()
    elif act == "swish":
        sg = torch.sigmoid(zf)
        da = sg * (1 + zf * (1 - sg))
    else:
        da = torch.ones_like(zf)
    dh = dy.float() @ w2.float()
    keep = torch.ones_like(zf) if p == 0 else ((dz_u.float() != 0) | (da == 0) | (dh == 0)).float()
    ref_dz = alpha * dh * da * keep / (1 - p)
    err = (dz.float() - ref_dz).abs().max() / ref_dz.abs().max()
    assert err < 1.5e-2, float(err)
    ref_dx = dz.float() @ w1.float()
    rel = (dxn.float() - ref_dx).norm() / ref_dx.norm()
    assert rel < 1e-2, float(rel)


@pytest.mark.parametrize("M,F,up,split", [(200, 256, False, None), (4096, 2048, True, None), (1000, 512, True, None),
                                          (16000, 2048, True, None), (8000, 2048, True, None), (4096, 2048, True, 2)])
def test_ffn_fused_bwd_layernorm_epilogue(M, F, up, split, ffn_split):
    """s2t_ffn_fused_bwd with ln_x: dx, its dropped copy and the dgamma / dbeta partial sums against s2t_layernorm_bwd run on
    the dxn the plain form of the kernel writes (modules/layer_norm.py:30-35 backward + the residual-branch gradient)."""
    from s2t_amd import functional as Fn

    if split is not None:
        ffn_split(split)
    d = 256
    g = torch.Generator().manual_seed(M)
    dy = torch.randn(M, d, generator=g).bfloat16().to(DEV)
    z = torch.randn(M, F, generator=g).bfloat16().to(DEV)
    w1t = (torch.randn(d, F, generator=g) * d ** -0.5).bfloat16().to(DEV)
    w2t = (torch.randn(F, d, generator=g) * F ** -0.5).bfloat16().to(DEV)
    x = (2 * torch.randn(M, d, generator=g) + 0.3).bfloat16().to(DEV)
    gam = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    dres = torch.randn(M, d, generator=g).bfloat16().to(DEV)
    xf = x.float()
    mean = xf.mean(1)
    rstd = (xf.var(1, unbiased=False) + 1e-5).rsqrt()
    Fn.DROPOUT.begin_step(torch.device(DEV))
    Fn.DROPOUT.set_seed(9)
    drop_h = Fn.DROPOUT.next(0.1, torch.device(DEV))
    up_drop = Fn.DROPOUT.next(0.1, torch.device(DEV)) if up else None
    dz0 = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    dxn = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    K.ffn_fused_bwd(dy, w2t, w1t, z, dz0, dxn, act="swish", alpha=0.5, drop_h=drop_h)
    dx_u = torch.empty_like(x)
    dxd_u = torch.empty_like(x) if up else None
    dg_u = torch.zeros(d, device=DEV)
    db_u = torch.zeros(d, device=DEV)
    K.layernorm_bwd(x, gam, dxn, mean, rstd, dx_u, dg_u, db_u, M, d, None, 0, dres, dx_drop=dxd_u, drop=up_drop)
    dz1 = torch.empty_like(dz0)
    dx = torch.full_like(x, 5.0)
    dxd = torch.full_like(x, 5.0) if up else None
    ws = torch.zeros(K.LN_REPLICAS * 2 * d, device=DEV)
    K.ffn_fused_bwd(dy, w2t, w1t, z, dz1, None, act="swish", alpha=0.5, drop_h=drop_h,
                    ln=dict(x=x, gamma=gam, mean=mean, rstd=rstd, ws=ws, dx=dx, dres=dres, dx_drop=dxd, drop=up_drop))
    torch.cuda.synchronize()
    K.ffn_exchange_check()
    assert torch.equal(dz0, dz1)
    rel = (dx.float() - dx_u.float()).norm() / dx_u.float().norm()
    assert rel < 6e-3, float(rel)  # (the epilogue works on fp32 dxn, the separate kernel on its bf16 rounding)
    w = ws.view(K.LN_REPLICAS, 2, d).sum(0)
    assert (w[0] - dg_u).abs().max() <= 1e-2 * dg_u.abs().max() + 1e-3
    assert (w[1] - db_u).abs().max() <= 1e-2 * db_u.abs().max() + 1e-3
    if up:
        # the dropped copy is the dropout of the STORED dx under the same mask
        keep = dxd_u.float() != 0
        assert ((dxd.float() != 0) == keep).float().mean() > 0.999
        ref = torch.where(keep, dx.float() / 0.9, torch.zeros_like(dx.float()))
        assert (dxd.float() - ref).abs().max() <= 2e-2 * ref.abs().max()


@pytest.mark.parametrize("M,Kd,mask,up", [(200, 256, False, False), (4000, 768, False, True), (4 * 250, 512, True, True)])
def test_rowblock_dgrad_matches_gemm_plus_layernorm_bwd(M, Kd, mask, up):
    """s2t_rowblock_dgrad (input gradient of the projection behind a LayerNorm + that LayerNorm's backward) against the dgrad
    GEMM followed by s2t_layernorm_bwd: dx, the dropped copy, the dgamma / dbeta partial sums; and its plain form (dxn)."""
    from s2t_amd import functional as Fn

    d = 256
    g = torch.Generator().manual_seed(M + Kd)
    dy = torch.randn(M, Kd, generator=g).bfloat16().to(DEV)
    w = (torch.randn(Kd, d, generator=g) * Kd ** -0.5).bfloat16().to(DEV)   # nn.Linear(d -> Kd) weight [Kd, d]
    wt = w.t().contiguous()                                                  # [256, Kd]
    x = (2 * torch.randn(M, d, generator=g) + 0.3).bfloat16().to(DEV)
    gam = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    dres = torch.randn(M, d, generator=g).bfloat16().to(DEV)
    xf = x.float()
    mean = xf.mean(1)
    rstd = (xf.var(1, unbiased=False) + 1e-5).rsqrt()
    T = 250
    lens = torch.tensor([250, 180, 37, 1][:M // T], dtype=torch.int32, device=DEV) if mask else None
    Fn.DROPOUT.begin_step(torch.device(DEV))
    Fn.DROPOUT.set_seed(4)
    up_drop = Fn.DROPOUT.next(0.1, torch.device(DEV)) if up else None
    # reference chain
    dxn_u = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    K.gemm(dy, w, dxn_u, M=M, N=d, K=Kd, lda=Kd, ldb=d, ldc=d, b_kmajor=True)
    dx_u = torch.empty_like(x)
    dxd_u = torch.empty_like(x) if up else None
    dg_u, db_u = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    K.layernorm_bwd(x, gam, dxn_u, mean, rstd, dx_u, dg_u, db_u, M, d, lens, T if mask else 0, dres, dx_drop=dxd_u, drop=up_drop)
    # plain form
    dxn = torch.full((M, d), 5.0, dtype=torch.bfloat16, device=DEV)
    K.rowblock_dgrad(dy, wt, dxn=dxn)
    # with the LayerNorm backward
    dx = torch.full_like(x, 5.0)
    dxd = torch.full_like(x, 5.0) if up else None
    ws = torch.zeros(K.LN_REPLICAS * 2 * d, device=DEV)
    K.rowblock_dgrad(dy, wt, ln=dict(x=x, gamma=gam, mean=mean, rstd=rstd, ws=ws, dx=dx, dres=dres, lens=lens, T=T if mask else 0,
                                     dx_drop=dxd, drop=up_drop))
    torch.cuda.synchronize()
    rel = (dxn.float() - dxn_u.float()).norm() / dxn_u.float().norm()
    assert rel < 4e-3, float(rel)
    rel = (dx.float() - dx_u.float()).norm() / dx_u.float().norm()
    assert rel < 6e-3, float(rel)
    wsum = ws.view(K.LN_REPLICAS, 2, d).sum(0)
    assert (wsum[0] - dg_u).abs().max() <= 1e-2 * dg_u.abs().max() + 1e-3
    assert (wsum[1] - db_u).abs().max() <= 1e-2 * db_u.abs().max() + 1e-3
    if up:
        keep = dxd_u.float() != 0
        assert ((dxd.float() != 0) == keep).float().mean() > 0.999
        ref = torch.where(keep, dx.float() / 0.9, torch.zeros_like(dx.float()))
        assert (dxd.float() - ref).abs().max() <= 2e-2 * ref.abs().max()


@pytest.mark.parametrize("end_norm,mask", [(False, False), (True, False), (True, True)])
def test_ffn_block_fused_vs_composed_training(end_norm, mask):
    """functional.ffn_block: the fused launch and the LayerNorm / GEMM composition give the same output and gradients
    (same dropout sites, so the same masks) — modules/s2t_transformer_layer.py:258-265, :311-320."""
    from s2t_amd import functional as Fn
    from s2t_amd import modules as Md
    from s2t_amd.flat_params import FlatParameters

    torch.manual_seed(5)
    d, F, B, T = 256, 512, 4, 40
    M = B * T

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.norm = Md.LayerNorm(d)
            self.ffn = Md.FeedForwardModule(d, F, 0.1, 0.1, "swish")
            self.end = Md.LayerNorm(d)
            self.pre = Md.Linear(d, d)

    blk = Blk().to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    flat = FlatParameters(blk, torch.bfloat16)
    blk.train()
    lens = torch.tensor([40, 33, 17, 5], dtype=torch.int32, device=DEV) if mask else None
    x0 = torch.randn(M, d, device=DEV).bfloat16()
    dy = torch.randn(M, d, device=DEV).bfloat16()
    outs = {}
    old = (Fn._FFN_FUSED, Fn._FFN_FUSED_MIN_ROWS)
    try:
        for fused in (True, False):
            Fn._FFN_FUSED, Fn._FFN_FUSED_MIN_ROWS = fused, 0
            Fn.DROPOUT.begin_step(DEV)
            Fn.DROPOUT.set_seed(77)
            flat.zero_grad()
            x = x0.clone().requires_grad_(True)
            xin = blk.pre(x)  # a producer in front, so that dx flows through a real graph
            y = blk.ffn.block(xin, blk.norm, 0.5, blk.end if end_norm else None, lens, T)
            y.backward(dy)
            torch.cuda.synchronize()
            outs[fused] = (y.detach().float(), x.grad.float().clone(), flat.grad.clone())
    finally:
        Fn._FFN_FUSED, Fn._FFN_FUSED_MIN_ROWS = old
    (yf, dxf, gf), (yc, dxc, gc) = outs[True], outs[False]

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    assert rel(yf, yc) < 1e-2, rel(yf, yc)
    assert rel(dxf, dxc) < 2e-2, rel(dxf, dxc)
    assert rel(gf, gc) < 2e-2, rel(gf, gc)


@pytest.mark.parametrize("M,N,act,use_ln,mask,res,p", [
    (64, 64, None, False, False, False, 0.0),
    (200, 768, None, True, False, False, 0.0),        # fused QKV projection behind self_attn_layer_norm, ragged last block
    (1000, 512, "glu", True, True, False, 0.0),       # conv_norm (padded rows zeroed) -> pointwise conv 1 -> GLU
    (1000, 256, None, False, True, True, 0.1),        # pointwise conv 2 / output projection: dropout, row mask, residual
    (16000, 768, None, True, False, False, 0.0),
    (16000, 512, "glu", True, True, False, 0.0),
    (16000, 256, None, False, False, True, 0.1),
])
def test_rowblock_gemm_matches_layernorm_plus_gemm(M, N, act, use_ln, mask, res, p):
    """s2t_rowblock_gemm against the kernels it replaces (s2t_layernorm_fwd + s2t_gemm with the same epilogue arguments and
    dropout site), which are themselves pinned to the oracle; plus the saves of the LayerNorm."""
    g = torch.Generator().manual_seed(M + N)
    d, T = 256, 50
    x = torch.randn(M, d, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
    bias = (0.1 * torch.randn(N, generator=g)).to(DEV) if act != "glu" else None
    gam = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    bet = (0.1 * torch.randn(d, generator=g)).to(DEV)
    nout = N // 2 if act == "glu" else N
    lens = None
    if mask and M % T == 0:
        lens = torch.randint(1, T + 1, (M // T,), generator=g).int().to(DEV)
    resid = torch.randn(M, nout, generator=g).bfloat16().to(DEV) if res else None
    seed = torch.tensor([4242], dtype=torch.int64, device=DEV)
    drop = (p, seed, 9) if p > 0 else None
    # reference path
    if use_ln:
        xl = torch.empty_like(x)
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        K.layernorm_fwd(x, gam, bet, xl, mean, rstd, M, d, 1e-5, lens, T)
    else:
        xl = x
    out_u = torch.empty(M, nout, dtype=torch.bfloat16, device=DEV)
    z_u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV) if act == "glu" else None
    K.gemm(xl, w, out_u, M=M, N=N, K=d, lda=d, ldb=d, ldc=nout, bias=bias, act=act, preact=z_u, ldp=N, alpha=0.5,
           residual=resid, ldr=nout, row_lens=lens if not use_ln else None, row_T=T, drop=drop)
    # fused
    out_f = torch.full((M, nout), float("nan"), dtype=torch.bfloat16, device=DEV)
    z_f = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV) if act == "glu" else None
    xl_f = torch.empty_like(x) if use_ln else None
    mean_f, rstd_f = (torch.empty(M, device=DEV), torch.empty(M, device=DEV)) if use_ln else (None, None)
    K.rowblock_gemm(x, w, out_f, N=N, ldc=nout, bias=bias, act=act, alpha=0.5, residual=resid, ldr=nout, preact=z_f, ldp=N,
                    ln=(gam, bet) if use_ln else None, ln_lens=lens if use_ln else None, ln_T=T, x_ln=xl_f,
                    ln_stats=(mean_f, rstd_f) if use_ln else None, row_lens=lens if not use_ln else None, row_T=T, drop=drop)
    torch.cuda.synchronize()
    if use_ln:
        assert float((mean_f - mean).abs().max()) < 1e-5 and float(((rstd_f - rstd) / rstd).abs().max()) < 1e-5
        assert float((xl_f.float() - xl.float()).abs().max()) <= 2 ** -5
    ou, of = out_u.float(), out_f.float()
    assert torch.isfinite(of).all()
    assert float((of - ou).abs().max() / ou.abs().max()) < 2e-2
    if p > 0:  # same masks: the dropped elements coincide (the residual shifts them away from zero: compare out - residual)
        du, df = (ou - resid.float()).abs() < 1e-6, (of - resid.float()).abs() < 1e-6
        assert float((du != df).float().mean()) < 2e-3
    if act == "glu":
        assert float((z_f.float() - z_u.float()).abs().max() / z_u.float().abs().max()) < 2e-2


@pytest.mark.parametrize("M,mask,res,p", [(250, True, True, 0.0), (1000, True, True, 0.1), (130, False, False, 0.0)])
def test_rowblock_gemm_affine_activation_prologue(M, mask, res, p):
    """pre = (scale, shift, act): the BatchNorm apply + activation + mask in the prologue must give what s2t_bn_act_fwd
    followed by the plain row-block projection gives — the staged tile (x_ln), the output and its dropout mask."""
    g = torch.Generator().manual_seed(M)
    d, T = 256, 125 if M % 125 == 0 else M
    B = M // T
    D = torch.randn(M, d, generator=g).bfloat16().to(DEV)
    w = (torch.randn(d, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
    scale = (1 + 0.2 * torch.randn(d, generator=g)).to(DEV)
    shift = (0.3 * torch.randn(d, generator=g)).to(DEV)
    lens = torch.tensor([T - 3 * b for b in range(B)], dtype=torch.int32, device=DEV) if mask else None
    resid = torch.randn(M, d, generator=g).bfloat16().to(DEV) if res else None
    seed = torch.tensor([77], dtype=torch.int64, device=DEV)
    drop = (p, seed, 9) if p > 0 else None
    a_ref = torch.empty_like(D)
    K.bn_act_fwd(D, a_ref, scale, shift, "swish", M, d, lens, T)
    y_ref = torch.empty_like(D)
    K.rowblock_gemm(a_ref, w, y_ref, N=d, ldc=d, residual=resid, ldr=d, row_lens=lens, row_T=T, drop=drop)
    a = torch.full_like(D, float("nan"))
    y = torch.empty_like(D)
    K.rowblock_gemm(D, w, y, N=d, ldc=d, residual=resid, ldr=d, row_lens=lens, row_T=T, drop=drop,
                    pre=(scale, shift, "swish"), ln_lens=lens, ln_T=T, x_ln=a)
    torch.cuda.synchronize()
    # the activation is evaluated by the same device function in both kernels; allow one bf16 ulp for contraction differences
    torch.testing.assert_close(a.float(), a_ref.float(), rtol=8e-3, atol=1e-3)
    torch.testing.assert_close(y.float(), y_ref.float(), rtol=2e-2, atol=2e-2)
    if drop is not None:  # the same elements are dropped (residual-only values survive there)
        base = resid.float() if res else torch.zeros_like(y.float())
        assert ((y.float() == base) == (y_ref.float() == base)).all()


@pytest.mark.parametrize("B,T,mask,res", [(5, 50, True, True), (64, 250, True, True), (3, 127, False, False), (7, 18, True, False)])
def test_rowblock_gemm_depthwise_conv_prologue(B, T, mask, res):
    """conv = (taps, T, running statistics): the eval-mode middle of the convolution module (15-tap depthwise conv over time
    with per-utterance zero padding, BatchNorm on the running statistics, swish, padded-frame mask) in pointwise conv 2's
    prologue, against (a) s2t_dwconv_bn_eval_fwd followed by the plain row-block projection and (b) torch's conv1d in fp32.
    T = 50 / 127 / 18 are not multiples of the 4-row groups, so groups straddle utterance boundaries; B*T is ragged against
    the 64-row blocks."""
    import torch.nn.functional as F
    g_ = torch.Generator().manual_seed(B * 1000 + T)
    d, Kw, M = 256, 15, B * T
    lens_h = torch.tensor([max(T - 4 * b, 9) for b in range(B)], dtype=torch.int32)
    G = torch.randn(B, T, d, generator=g_)
    if mask:  # the module's input is zero on padded frames, and pointwise conv 1 + GLU keep it so
        G = G * (torch.arange(T)[None, :, None] < lens_h[:, None, None])
    G = G.bfloat16().reshape(M, d).to(DEV)
    wd = (torch.randn(d, Kw, generator=g_) * 0.3).to(DEV)
    w = (torch.randn(d, d, generator=g_) * d ** -0.5).bfloat16().to(DEV)
    gamma = (1 + 0.2 * torch.randn(d, generator=g_)).to(DEV)
    beta = (0.3 * torch.randn(d, generator=g_)).to(DEV)
    rmean = (0.2 * torch.randn(d, generator=g_)).to(DEV)
    rvar = (0.5 + torch.rand(d, generator=g_)).to(DEV)
    lens = lens_h.to(DEV) if mask else None
    resid = torch.randn(M, d, generator=g_).bfloat16().to(DEV) if res else None
    a_ref = torch.empty_like(G)
    K.dwconv_bn_eval_fwd(G, wd, a_ref, B, T, d, Kw, gamma, beta, rmean, rvar, 1e-5, "swish", lens)
    y_ref = torch.empty_like(G)
    K.rowblock_gemm(a_ref, w, y_ref, N=d, ldc=d, residual=resid, ldr=d, row_lens=lens, row_T=T)
    y = torch.empty_like(G)
    K.rowblock_gemm(G, w, y, N=d, ldc=d, residual=resid, ldr=d, row_lens=lens, row_T=T, pre=(gamma, beta, "swish"),
                    ln_lens=lens, ln_T=T, conv=(wd, T, rmean, rvar, 1e-5))
    torch.cuda.synchronize()
    torch.testing.assert_close(y.float(), y_ref.float(), rtol=2e-2, atol=3e-2)
    # (b) fp32 torch: conv1d (groups = d, pad 7) -> BatchNorm eval -> swish -> mask -> bf16 -> projection (+ mask, residual)
    x = G.float().view(B, T, d).transpose(1, 2)
    c = F.conv1d(x, wd.view(d, 1, Kw), padding=7, groups=d).transpose(1, 2)
    c = (c - rmean) * torch.rsqrt(rvar + 1e-5) * gamma + beta
    c = c * torch.sigmoid(c)
    keep = (torch.arange(T, device=DEV)[None, :] < lens[:, None]) if mask else torch.ones(B, T, dtype=torch.bool, device=DEV)
    c = (c * keep[:, :, None]).bfloat16().float().reshape(M, d)
    yy = (c @ w.float().t()) * keep.reshape(M, 1)
    if res:
        yy = yy + resid.float()
    torch.testing.assert_close(y.float(), yy, rtol=2e-2, atol=3e-2)


def test_ffn_split_follows_the_row_count(ffn_split):
    """Workgroups per 128-row block of the fused feed-forward kernels (csrc/ffn_pc.hip), as the launch itself reports it
    (s2t_ffn_fused_describe prints the kernel name rocprofv3 shows): eight parts for the decoder's few thousand rows, two for
    the encoder's 16 000, one when pinned; the result is the same function of its inputs in every form."""
    import ctypes as C
    from s2t_amd import _lib as L
    d, F = 256, 2048
    g = torch.Generator().manual_seed(11)
    w1 = (torch.randn(F, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
    w2 = (torch.randn(d, F, generator=g) * F ** -0.5).bfloat16().to(DEV)
    b1 = (0.1 * torch.randn(F, generator=g)).to(DEV)
    b2 = (0.1 * torch.randn(d, generator=g)).to(DEV)
    gam, bet = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)

    def run(M):
        x = torch.randn(M, d, generator=torch.Generator().manual_seed(M)).bfloat16().to(DEV)
        y = torch.empty_like(x)
        K.GEMM_PROFILE = []
        try:
            K.ffn_fused_fwd(x, w1, b1, w2, b2, y, act="relu", alpha=1.0, residual=x, ln=(gam, bet))
            torch.cuda.synchronize()
            name = K.GEMM_PROFILE[0][0]
        finally:
            K.GEMM_PROFILE = None
        return name, y.float()

    n8, y8 = run(3904)
    assert n8.endswith(", 8>"), n8
    n2, _ = run(16000)
    assert n2.endswith(", 2>"), n2
    n4, y4 = run(8000)
    assert n4.endswith(", 4>"), n4
    ffn_split(1)
    _, y4_1 = run(8000)
    ffn_split(0)
    assert float((y4 - y4_1).abs().max()) <= 2e-2 * float(y4_1.abs().max())
    ffn_split(2)
    m2, y2 = run(3904)
    assert m2.endswith(", 2>"), m2
    ffn_split(1)
    m1, y1 = run(3904)
    assert m1.endswith(", 1>"), m1
    # same products, added in a different order: fp32 partial sums, one bf16 rounding at the end
    assert float((y8 - y1).abs().max()) <= 2e-2 * float(y1.abs().max())
    assert float((y2 - y1).abs().max()) <= 2e-2 * float(y1.abs().max())


@pytest.mark.parametrize("M", [16000, 3904])
def test_ffn_exchange_timeout_is_reported(M):
    """A partner workgroup whose flag never arrives (the s2t_ffn_configure test hook: part 1 of every row block keeps its flag
    down, short spin limit) must not pass silently: the launch counts the event in the workspace's error word, the host
    check raises and re-zeroes the flag area, and the next launch — hook off — runs clean and gives the pinned form's result."""
    d, F = 256, 2048
    x, w1, b1, w2, b2, gam, bet, eg, eb = _mk(M, F, 5)
    xd, w1d, b1d, w2d, b2d, gd, bd = (t.to(DEV) for t in (x, w1, b1, w2, b2, gam, bet))

    def run():
        y = torch.empty_like(xd)
        K.ffn_fused_fwd(xd, w1d, b1d, w2d, b2d, y, act="swish", alpha=0.5, residual=xd, ln=(gd, bd))
        torch.cuda.synchronize()
        return y.float()

    y_ok = run()
    K.ffn_exchange_check()  # clean
    K.ffn_configure(fault=1)
    try:
        run()
        with pytest.raises(RuntimeError, match="timed out"):
            K.ffn_exchange_check()
        # the non-blocking form the Trainer uses: queue, let the copy land, examine
        run()
        K.ffn_exchange_poll()
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="timed out"):
            K.ffn_exchange_poll()
    finally:
        K.ffn_configure(fault=0)
    y_again = run()
    K.ffn_exchange_check()
    assert torch.equal(y_again, y_ok)


@pytest.mark.parametrize("parked", [32, 64])
def test_split_ffn_forms_with_compute_units_held_by_another_kernel(parked, ffn_split):
    """The parts of a row block wait for each other's partial rows (csrc/ffn_pc.hip), and their number is sized for a grid that is
    resident at once.  In a data-parallel step RCCL's all-reduce kernels run beside backward (legacy_distributed_data_parallel.py:
    76-160) and take compute units.  Rehearsal on one GPU: a kernel on a side stream holds 32 / 64 CUs (s2t_occupy_cus: one
    workgroup with 96 KiB of LDS each, so no 128 - 160 KiB workgroup of the fused kernels fits beside it) while the two-part
    (16 000 rows, training forward and backward shapes) and eight-part (3 904 rows) forms run:
      * no exchange time-out (the partners of a resident workgroup are dispatched within 64 blocks of it and the grid drains in
        dispatch order: earlier groups finish and free their CUs), results equal to the one-workgroup form's;
      * with the CU budget lowered to what is free (s2t_ffn_cu_budget, what the data-parallel wrapper does) the launch picks a
        split whose grid fits again;
      * the slowdown is printed."""
    import time

    d, F = 256, 2048
    side = torch.cuda.Stream()
    stop = torch.zeros(1, dtype=torch.int32, device=DEV)
    arrived = torch.zeros(1, dtype=torch.int32, device=DEV)
    cus = K.ffn_cu_budget()
    res = {}
    for M in (16000, 3904):
        x, w1, b1, w2, b2, gam, bet, eg, eb = _mk(M, F, 9)
        xd, w1d, b1d, w2d, b2d, gd, bd = (t.to(DEV) for t in (x, w1, b1, w2, b2, gam, bet))

        def run(n=1):
            y = torch.empty_like(xd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                K.ffn_fused_fwd(xd, w1d, b1d, w2d, b2d, y, act="swish", alpha=0.5, residual=xd, ln=(gd, bd))
            torch.cuda.current_stream().synchronize()
            return y.float(), (time.perf_counter() - t0) / n

        ffn_split(1)
        y1, _ = run()
        ffn_split(0)
        run(3)
        y_free, t_free = run(20)
        stop.zero_()
        arrived.zero_()
        K.occupy_cus(parked, 1500, stop, arrived, stream=side)
        while int(arrived.cpu()) < parked:  # the occupier's workgroups are resident
            time.sleep(0.001)
        try:
            y_held, t_held = run(20)
            K.ffn_exchange_check()  # raises on an exchange time-out
            K.ffn_cu_budget(cus - parked)
            y_budget, t_budget = run(20)
            K.ffn_exchange_check()
        finally:
            K.ffn_cu_budget(0)
            stop.fill_(1)
            side.synchronize()
        tol = 2e-2 * float(y1.abs().max())  # same products, another order of the fp32 partial sums, one bf16 rounding
        assert float((y_free - y1).abs().max()) <= tol
        assert torch.equal(y_held, y_free)  # the same split form: where a workgroup runs does not enter its arithmetic
        assert float((y_budget - y1).abs().max()) <= tol
        res[M] = (t_free * 1e6, t_held * 1e6, t_budget * 1e6)
    print("fused FFN eval forward with %d of %d CUs held: rows 16000 free %.1f us, held %.1f, held + budget %.1f; rows 3904 free %.1f, "
          "held %.1f, held + budget %.1f" % ((parked, cus) + res[16000] + res[3904]))


@pytest.mark.parametrize("M,packed,train,p,glu", [(16000, True, True, 0.1, True), (16000, False, True, 0.1, True),
                                                   (4033, True, False, 0.0, True), (250, False, True, 0.0, True),
                                                   (6000, True, True, 0.1, False)])
def test_rowblock_chain_equals_two_launches(M, packed, train, p, glu):
    """s2t_rowblock_chain (the attention output projection, then conv_norm + pointwise conv 1 + GLU on the rows it left in the
    workgroup — s2t_transformer_layer.py:283-288, convolution.py:86-92) against the two s2t_rowblock_gemm launches it
    replaces: every output of both stages bit for bit — uniform and packed rows, with and without dropout / training saves."""
    from s2t_amd import rows as Rows

    d, T = 256, 250
    B = (M + T - 1) // T
    Mr = B * T
    g = torch.Generator().manual_seed(M + 7 * packed + 3 * train)
    O = torch.randn(Mr, d, generator=g).bfloat16().to(DEV)
    res = torch.randn(Mr, d, generator=g).bfloat16().to(DEV)
    wo = (torch.randn(d, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
    bo = torch.randn(d, generator=g).to(DEV) * 0.1
    N2 = 2 * d if glu else 3 * d
    w1 = (torch.randn(N2, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
    b1 = torch.randn(N2, generator=g).to(DEV) * 0.1
    gam = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    bet = (0.1 * torch.randn(d, generator=g)).to(DEV)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g).to(torch.int32)
    lens[0] = T
    lens = lens.to(DEV)
    if packed:
        lens = Rows.attach(lens, B, T, 7)
    else:
        lens = Rows.detached(lens)
    seed = torch.tensor([99], dtype=torch.int64, device=DEV)
    nout = d if glu else N2

    def outs():
        return dict(y=torch.zeros(Mr, d, dtype=torch.bfloat16, device=DEV), o2=torch.zeros(Mr, nout, dtype=torch.bfloat16, device=DEV),
                    z=torch.zeros(Mr, N2, dtype=torch.bfloat16, device=DEV) if (train and glu) else None,
                    xl=torch.zeros(Mr, d, dtype=torch.bfloat16, device=DEV) if train else None,
                    mean=torch.zeros(Mr, device=DEV) if train else None, rstd=torch.zeros(Mr, device=DEV) if train else None)

    def stages(o):
        first = dict(x=O, w=wo, out=o["y"], N=d, ldc=d, bias=bo, residual=res, ldr=d, drop=(p, seed, 5) if p > 0 else None,
                     rows=lens if packed else None)
        second = dict(x=o["y"], w=w1, out=o["o2"], N=N2, ldc=nout, bias=b1, act="glu" if glu else None, preact=o["z"],
                      ldp=N2 if o["z"] is not None else 0, ln=(gam, bet), ln_lens=lens, ln_T=T, x_ln=o["xl"],
                      ln_stats=(o["mean"], o["rstd"]) if train else None)
        return first, second

    a, b = outs(), outs()
    f, s = stages(a)
    K.rowblock_gemm(f.pop("x"), f.pop("w"), f.pop("out"), **f)
    K.rowblock_gemm(s.pop("x"), s.pop("w"), s.pop("out"), **s)
    f, s = stages(b)
    K.rowblock_chain(f, s)
    torch.cuda.synchronize()
    for k in a:
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), k
    assert float(a["o2"].float().abs().sum()) > 0
