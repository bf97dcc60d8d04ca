"""Entry points of include/s2t_hip.h on PACKED rows ("Packed rows" at the top of the header) against the same entry points on
the padded layout: the same values in both layouts give the same results on every row that holds a frame, nothing is read or
written beyond the live rows, and sums over rows (BatchNorm statistics, LayerNorm / bias / weight gradients, losses) see exactly
the frames (and halo rows) of the padded batch.  The padded layout is what tests/test_kernels_gpu.py pins against fp64 maths and
the oracle.  (Fused attention: tests/test_attn_fused_gpu.py; whole models: tests/test_packed_rows_gpu.py.)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from s2t_amd import kernels as K  # noqa: E402
from s2t_amd import rows as Rows  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def _geom(lens, T, halo):
    l32 = torch.tensor(lens, dtype=torch.int32, device=DEV)
    Rows.attach(l32, len(lens), T, halo)
    return l32, Rows.detached(l32)


def _valid(lens, T, extra=0):
    """[B*T] bool: frames (+ ``extra`` halo rows behind each utterance, never beyond T)."""
    l = torch.tensor(lens, device=DEV)
    return (torch.arange(T, device=DEV)[None, :] < (l + extra).clamp(max=T)[:, None]).reshape(-1)


def test_gemm_and_layernorm_stop_at_the_live_rows():
    lens, T = [40, 33, 7, 1], 40
    B = len(lens)
    lp, lu = _geom(lens, T, 3)
    live = lp._pk.live_rows()
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(B * T, 256, generator=g)).to(BF).to(DEV)
    w = (torch.randn(384, 256, generator=g) * 0.1).to(BF).to(DEV)
    bias = torch.randn(384, generator=g).to(DEV)
    xp = Rows.pack(x, lp)
    # ---- s2t_gemm: bound only (rows=) and bound + mask (row_lens=)
    for masked in (False, True):
        ref = torch.full((B * T, 384), 9.0, dtype=BF, device=DEV)
        out = torch.full((B * T, 384), 9.0, dtype=BF, device=DEV)
        K.gemm(x, w, ref, M=B * T, N=384, K=256, lda=256, ldb=256, ldc=384, bias=bias, row_lens=lu if masked else None, row_T=T)
        K.gemm(xp, w, out, M=B * T, N=384, K=256, lda=256, ldb=256, ldc=384, bias=bias, row_lens=lp if masked else None,
               rows=None if masked else lp)
        torch.cuda.synchronize()
        assert float(out[live:].float().min()) == 9.0 and float(out[live:].float().max()) == 9.0, "rows beyond the live ones written"
        v = _valid(lens, T)
        assert torch.equal(Rows.unpack(out, lp)[v], ref[v])
        if masked:  # halo rows are masked like padded frames
            m = lp._pk.row_map[:live]
            assert float(out[:live][m < 0].float().abs().max()) == 0.0
    # ---- LayerNorm forward / backward with the mask: dgamma / dbeta sum the frames only
    gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(256, generator=g)).to(DEV)
    dy = torch.randn(B * T, 256, generator=g).to(BF).to(DEV)
    dyp = Rows.pack(dy, lp)
    res = {}
    for name, (xx, dd, ll, tt) in {"padded": (x, dy, lu, T), "packed": (xp, dyp, lp, T)}.items():
        y = torch.full_like(xx, 5.0)
        mean = torch.zeros(B * T, device=DEV)
        rstd = torch.zeros(B * T, device=DEV)
        K.layernorm_fwd(xx, gamma, beta, y, mean, rstd, B * T, 256, 1e-5, ll, tt)
        dx = torch.full_like(xx, 5.0)
        dg = torch.zeros(256, device=DEV)
        db = torch.zeros(256, device=DEV)
        K.layernorm_bwd(xx, gamma, dd, mean, rstd, dx, dg, db, B * T, 256, ll, tt)
        torch.cuda.synchronize()
        res[name] = (y, dx, dg, db)
    v = _valid(lens, T)
    assert torch.equal(Rows.unpack(res["packed"][0], lp)[v], res["padded"][0][v])
    assert torch.equal(Rows.unpack(res["packed"][1], lp)[v], res["padded"][1][v])
    assert float(res["packed"][0][live:].float().min()) == 5.0  # untouched
    np.testing.assert_allclose(res["packed"][2].cpu().numpy(), res["padded"][2].cpu().numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(res["packed"][3].cpu().numpy(), res["padded"][3].cpu().numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("Kw", [15, 31])
def test_depthwise_conv_batchnorm_statistics_and_fused_backward(Kw):
    """modules/convolution.py:94-104 on packed rows: the depthwise convolution's outputs on frames AND halo rows, the BatchNorm
    batch statistics (all B * T positions of the padded batch: the halo rows are the only padded frames whose output is not zero),
    and the fused backward of the module's middle (dZ on frames, the depthwise weight gradient)."""
    lens, T, C = [64, 64, 57, 50, 31, 12, 3, 1], 64, 256
    B, pad = len(lens), (Kw - 1) // 2
    lp, lu = _geom(lens, T, pad)
    g = torch.Generator().manual_seed(Kw)
    fr = _valid(lens, T)
    G = torch.randn(B * T, C, generator=g).to(BF).to(DEV)
    G[~fr] = 0  # the GLU output is zero on padded frames (masked input, no bias)
    w = (torch.randn(C, Kw, generator=g) * 0.3).to(DEV)
    Gp = Rows.pack(G, lp)
    out = {}
    for name, (gg, ll) in {"padded": (G, lu), "packed": (Gp, lp)}.items():
        D = torch.zeros(B * T, C, dtype=BF, device=DEV)
        stats = torch.zeros(K.dwconv_stat_partials(B, T), 2, C, device=DEV)
        K.dwconv_fwd(gg, w, D, B, T, C, Kw, stats=stats, lens=ll if name == "packed" else None)
        torch.cuda.synchronize()
        out[name] = (D, stats.sum(0))
    both = _valid(lens, T, pad)  # frames + halo rows
    Dp = Rows.unpack(out["packed"][0], lp)  # (unpack moves frames only: halo rows are compared through the statistics)
    assert torch.equal(Dp[fr], out["padded"][0][fr])
    # padded frames beyond the halo are exactly zero in the padded layout: both layouts sum the same values
    assert float(out["padded"][0][~both].float().abs().max() if bool((~both).any()) else 0.0) == 0.0
    np.testing.assert_allclose(out["packed"][1].cpu().numpy(), out["padded"][1].cpu().numpy(), rtol=2e-5, atol=2e-3)
    # ---- fused backward of the middle
    count = float(B * T)
    mean = (out["padded"][1][0] / count)
    var = (out["padded"][1][1] / count - mean * mean).clamp_min(0)
    rstd = torch.rsqrt(var + 1e-5)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    scale = gamma * rstd
    shift = beta - mean * scale
    Z = torch.randn(B * T, 2 * C, generator=g).to(BF).to(DEV)
    dA = torch.randn(B * T, C, generator=g).to(BF).to(DEV)
    dA[~fr] = 0
    res = {}
    for name, ll in {"padded": lu, "packed": lp}.items():
        pk = name == "packed"
        Dd = out[name][0]
        dd, gg, zz = (Rows.pack(dA, lp), Gp, Rows.pack(Z, lp)) if pk else (dA, G, Z)
        sums = torch.zeros(2 * C, device=DEV)
        K.bn_act_bwd(Dd, dd, None, scale, shift, mean, rstd, sums, count, "swish", B * T, C, ll, T)
        dZ = torch.zeros(B * T, 2 * C, dtype=BF, device=DEV)
        dw = torch.zeros(C, Kw, device=DEV)
        K.conv_bwd_fused(Dd, dd, gg, zz, w, scale, shift, mean, rstd, sums, count, "swish", ll, dZ, dw, B, T, C, Kw)
        torch.cuda.synchronize()
        res[name] = (sums, Rows.unpack(dZ, lp) if pk else dZ, dw)
    np.testing.assert_allclose(res["packed"][0].cpu().numpy(), res["padded"][0].cpu().numpy(), rtol=1e-4, atol=1e-2)
    a, b = res["padded"][1][fr].float(), res["packed"][1][fr].float()
    assert float((a - b).norm() / a.norm()) < 2e-3  # (the folded BatchNorm sums differ in their last bits)
    np.testing.assert_allclose(res["packed"][2].cpu().numpy(), res["padded"][2].cpu().numpy(), rtol=2e-3, atol=3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_ctc_loss_gradient_and_greedy_collapse(dtype):
    """criterions/ctc.py:435-474 and s2t_ctc.py:312-347 on packed logit rows: the per-utterance nll, the gradient on every frame
    (zero on halo rows), the collapsed greedy ids."""
    lens, T, V, S = [50, 44, 30, 17, 9], 50, 37, 6
    B = len(lens)
    lp, lu = _geom(lens, T, 4)
    live = lp._pk.live_rows()
    g = torch.Generator().manual_seed(5)
    lbuf = torch.nn.functional.pad(torch.randn(B * T, V, generator=g), (0, 3)).to(dtype).to(DEV).contiguous()
    logits = lbuf[:, :V]                 # (rows of 40 elements in both layouts: the same vector path through a row)
    lpk = Rows.pack(lbuf, lp)[:, :V]
    tgt = torch.randint(1, V, (B, S), generator=g).to(DEV)
    tl = torch.tensor([6, 5, 6, 3, 2], dtype=torch.int32, device=DEV)
    Lmax = 2 * S + 1
    res = {}
    for name, (lg, ll, rr) in {"padded": (logits, lu, None), "packed": (lpk, lp, lp)}.items():
        lse = torch.zeros(B * T, device=DEV)
        K.argmax_lse(lg, lg.stride(0), B * T, V, None, None, lse, bound=rr)
        alpha = torch.zeros(B, T, Lmax, device=DEV)
        beta = torch.zeros(B, T, Lmax, device=DEV)
        nll = torch.zeros(B, device=DEV)
        K.ctc_loss_fwd(lg, lg.stride(0), B, T, V, lse, tgt, S, tl, ll, 0, alpha, beta, Lmax, nll, rows=rr)
        grad = torch.full((B * T, 40), 3.0, dtype=dtype, device=DEV)[:, :V]
        K.ctc_loss_bwd(lg, lg.stride(0), B, T, V, lse, tgt, S, tl, ll, 0, alpha, beta, Lmax, nll, 1.0, grad, grad.stride(0), rows=rr)
        idx = torch.zeros(B * T, dtype=torch.int32, device=DEV)
        top = torch.zeros(B * T, device=DEV)
        K.argmax_lse(lg, lg.stride(0), B * T, V, idx, top, None, bound=rr)
        toks = torch.zeros(B, T, dtype=torch.int64, device=DEV)
        olen = torch.zeros(B, dtype=torch.int32, device=DEV)
        osc = torch.zeros(B, device=DEV)
        K.ctc_collapse(idx, top, ll, B, T, 0, toks, olen, osc, rows=rr)
        torch.cuda.synchronize()
        res[name] = (nll, grad, toks, olen)
    assert torch.equal(res["padded"][0], res["packed"][0])
    fr = _valid(lens, T)
    gp = torch.zeros(B * T, 40, dtype=dtype, device=DEV)
    K.pack_rows(torch.nn.functional.pad(res["packed"][1], (0, 3)).contiguous(), gp, lp, False)
    assert torch.equal(gp[:, :V][fr], res["padded"][1][fr])
    m = lp._pk.row_map[:live]
    assert float(res["packed"][1][:live][m < 0].float().abs().max()) == 0.0  # halo rows: zero gradient
    assert float(res["packed"][1][live:].float().min()) == 3.0                # beyond the live rows: untouched
    assert torch.equal(res["padded"][3], res["packed"][3])
    for b in range(B):
        n = int(res["padded"][3][b])
        assert res["padded"][2][b, :n].tolist() == res["packed"][2][b, :n].tolist()
