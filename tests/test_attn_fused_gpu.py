"""Fused attention kernels (bf16, head dim 64) vs a float64 CPU attention on the bf16-rounded operands."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from s2t_amd import kernels as K  # noqa: E402

DEV = "cuda"


def ref_attention(q, k, v, klen, causal, scale, p=None, u=None, vb=None):
    """q,k,v: (B,T,H,dk) float64; returns O (B,Tq,H,dk), lse (B,H,Tq), probabilities."""
    B, Tq, H, dk = q.shape
    Tk = k.shape[1]
    qh, kh, vh = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)
    if p is not None:
        # the kernel rounds q+u / q+v to bf16 (as the composed path does)
        qu = (qh + u[None, :, None, :]).to(torch.bfloat16).double()
        qv = (qh + vb[None, :, None, :]).to(torch.bfloat16).double()
        ac = qu @ kh.transpose(-1, -2)
        bd_full = qv @ p.permute(1, 2, 0)[None]  # p: (2T-1, H, dk) -> (H, dk, 2T-1)
        idx = (Tq - 1) - torch.arange(Tq)[:, None] + torch.arange(Tk)[None, :]
        bd = torch.gather(bd_full, 3, idx[None, None].expand(B, H, Tq, Tk))
        s = (ac + bd) * scale
    else:
        s = (qh @ kh.transpose(-1, -2)) * scale
    mask = torch.arange(Tk)[None, :] >= klen[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    if causal:
        s = s + torch.triu(torch.full((Tq, Tk), float("-inf"), dtype=torch.float64), 1)
    lse = torch.logsumexp(s, -1)
    pr = torch.softmax(s, -1)
    o = (pr @ vh).permute(0, 2, 1, 3)
    return o, lse, pr


@pytest.mark.parametrize("Tq,Tk,causal,rel", [(250, 250, False, False), (61, 61, True, False), (61, 250, False, False),
                                              (250, 250, False, True), (100, 100, False, True), (17, 17, False, True),
                                              # BASELINE config 3 (PDS Conformer, 64 x 2000): stage lengths 1004 / 502 / 251
                                              (1004, 1004, False, True), (502, 502, False, True), (251, 251, False, True)])
def test_fused_forward(Tq, Tk, causal, rel):
    g = torch.Generator().manual_seed(Tq * 7 + Tk + rel)
    B, H, dk = 3, 4, 64
    d = H * dk
    bf = torch.bfloat16
    qkv = (torch.randn(B, Tq, 3 * d, generator=g) * 0.7).to(bf)
    if Tq == Tk:
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        qd = kd = vd = qkv.to(DEV)
        qs = (Tq * 3 * d, 3 * d)
        qp, kp, vp = qd, qd[..., d:], qd[..., 2 * d:]
        ks = vs = qs
    else:
        q = qkv[..., :d].contiguous()
        kv = (torch.randn(B, Tk, 2 * d, generator=g) * 0.7).to(bf)
        k, v = kv[..., :d], kv[..., d:]
        qp = q.to(DEV)
        kvd = kv.to(DEV)
        kp, vp = kvd, kvd[..., d:]
        qs, ks, vs = (Tq * d, d), (Tk * 2 * d, 2 * d), (Tk * 2 * d, 2 * d)
    klen = torch.tensor([Tk, max(1, Tk - 7), max(1, Tk // 2)], dtype=torch.int32)
    scale = 1.0 / math.sqrt(dk)
    o = torch.full((B, Tq, d), 9.0, dtype=bf, device=DEV)
    lse = torch.empty(B * H, Tq, dtype=torch.float32, device=DEV)
    pos = u = vb = None
    pp = None
    if rel:
        pos = (torch.randn(2 * Tq - 1, d, generator=g) * 0.7).to(bf)
        u = torch.randn(H, dk, generator=g) * 0.3
        vb = torch.randn(H, dk, generator=g) * 0.3
        pp = pos.to(DEV)
    K.attn_fused_fwd(qp, qs[0], qs[1], kp, ks[0], ks[1], vp, vs[0], vs[1], o, Tq * d, d, lse, B, H, Tq, Tk, dk,
                     klen.to(DEV), causal, scale, pp, d if rel else 0, u.reshape(-1).to(DEV) if rel else None,
                     vb.reshape(-1).to(DEV) if rel else None)
    torch.cuda.synchronize()
    oref, lref, _ = ref_attention(q.double().view(B, Tq, H, dk), k.double().view(B, Tk, H, dk), v.double().view(B, Tk, H, dk),
                                  klen.long(), causal, scale, pos.double().view(-1, H, dk) if rel else None,
                                  u.double() if rel else None, vb.double() if rel else None)
    got = o.cpu().double().view(B, Tq, H, dk)
    # P is rounded to bf16 before multiplying V and O is stored in bf16
    np.testing.assert_allclose(got.numpy(), oref.numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(lse.cpu().double().view(B, H, Tq).numpy(), lref.numpy(), rtol=1e-3, atol=2e-3)


def _dropout_mask(Z, Tq, Tk, drop):
    """Keep-mask of the attention-dropout site, read back through the composed path's softmax kernel (same hash of
    ((z*Tq+i)*Tk+j)): uniform probabilities in, dropped probabilities out."""
    ld = (Tk + 7) // 8 * 8
    S = torch.zeros(Z, Tq, ld, dtype=torch.float32, device=DEV)
    P = torch.empty(Z, Tq, ld, dtype=torch.float32, device=DEV)
    Pd = torch.empty(Z, Tq, ld, dtype=torch.float32, device=DEV)
    K.attn_softmax_fwd(S, ld, None, 0, P, ld, Z, 1, Tq, Tk, 1.0, None, False, False, Pd, drop)
    torch.cuda.synchronize()
    return (Pd[:, :, :Tk] != 0).cpu()


@pytest.mark.parametrize("Tq,Tk,causal,rel,pdrop", [(250, 250, False, False, 0.0), (61, 61, True, False, 0.0),
                                                    (61, 250, False, False, 0.0), (130, 130, False, False, 0.1),
                                                    (250, 250, False, True, 0.0), (100, 100, False, True, 0.1),
                                                    (17, 17, False, True, 0.0),
                                                    (1004, 1004, False, True, 0.0), (502, 502, False, True, 0.0),
                                                    (251, 251, False, True, 0.0)])
def test_fused_backward(Tq, Tk, causal, rel, pdrop):
    g = torch.Generator().manual_seed(Tq * 11 + Tk + rel)
    B, H, dk = 3, 4, 64
    d = H * dk
    Z = B * H
    bf = torch.bfloat16
    q = (torch.randn(B, Tq, d, generator=g) * 0.7).to(bf)
    k = (torch.randn(B, Tk, d, generator=g) * 0.7).to(bf)
    v = (torch.randn(B, Tk, d, generator=g) * 0.7).to(bf)
    dO = (torch.randn(B, Tq, d, generator=g) * 0.5).to(bf)
    klen = torch.tensor([Tk, max(1, Tk - 7), max(1, Tk // 2)], dtype=torch.int32)
    scale = 1.0 / math.sqrt(dk)
    pos = u = vb = None
    if rel:
        pos = (torch.randn(2 * Tq - 1, d, generator=g) * 0.7).to(bf)
        u = torch.randn(H, dk, generator=g) * 0.3
        vb = torch.randn(H, dk, generator=g) * 0.3
    seed = torch.full((1,), 1234, dtype=torch.int64, device=DEV)
    drop = (pdrop, seed, 5) if pdrop > 0 else None
    qd, kd, vd, dOd = q.to(DEV), k.to(DEV), v.to(DEV), dO.to(DEV)
    o = torch.empty(B, Tq, d, dtype=bf, device=DEV)
    lse = torch.empty(Z, Tq, dtype=torch.float32, device=DEV)
    pp = pos.to(DEV) if rel else None
    ud = u.reshape(-1).to(DEV) if rel else None
    vbd = vb.reshape(-1).to(DEV) if rel else None
    kl = klen.to(DEV)
    K.attn_fused_fwd(qd, Tq * d, d, kd, Tk * d, d, vd, Tk * d, d, o, Tq * d, d, lse, B, H, Tq, Tk, dk, kl, causal, scale, pp,
                     d if rel else 0, ud, vbd, drop)
    delta = torch.empty(Z, Tq, dtype=torch.float32, device=DEV)
    dq = torch.full((B, Tq, d), 7.0, dtype=bf, device=DEV)
    dkk = torch.full((B, Tk, d), 7.0, dtype=bf, device=DEV)
    dv = torch.full((B, Tk, d), 7.0, dtype=bf, device=DEV)
    ldb = (2 * Tq - 1 + 7) // 8 * 8
    dbd = torch.full((H, B, Tq, ldb), 7.0, dtype=bf, device=DEV) if rel else None
    K.attn_fused_bwd(qd, Tq * d, d, kd, Tk * d, d, vd, Tk * d, d, o, dOd, Tq * d, d, lse, delta, dq, dkk, dv, dbd, ldb, B, H,
                     Tq, Tk, dk, kl, causal, scale, pp, d if rel else 0, ud, vbd, drop)
    torch.cuda.synchronize()
    if rel:
        # dbd_band_only: into a buffer that is already zero outside the band, twice (the second call overwrites the band)
        dbd2 = torch.zeros_like(dbd)
        qv_out = torch.full((B * Tq, d), 9.0, dtype=bf, device=DEV)
        for _ in range(2):
            K.attn_fused_bwd(qd, Tq * d, d, kd, Tk * d, d, vd, Tk * d, d, o, dOd, Tq * d, d, lse, torch.empty_like(delta),
                             torch.empty_like(dq), torch.empty_like(dkk), torch.empty_like(dv), dbd2, ldb, B, H, Tq, Tk, dk, kl,
                             causal, scale, pp, d, ud, vbd, drop, dbd_band_only=True, qv_out=qv_out)
        torch.cuda.synchronize()
        assert torch.equal(dbd2[..., :2 * Tq - 1], dbd[..., :2 * Tq - 1])
        # qv_out: q + pos_bias_v as the kernel rounds it (the operand of the position-table gradient)
        ref_qv = (qd.float().view(B * Tq, d) + vbd.view(1, d)).to(bf)
        assert torch.equal(qv_out, ref_qv)
    # delta is an output of the dQ kernel
    ref_delta = (dOd.float().view(B, Tq, H, dk) * o.float().view(B, Tq, H, dk)).sum(-1).permute(0, 2, 1).reshape(Z, Tq)
    assert (delta - ref_delta).abs().max() <= 1e-3 * ref_delta.abs().max() + 1e-5

    # ---- float64 autograd reference on the same bf16-rounded operands
    qh = q.double().view(B, Tq, H, dk).permute(0, 2, 1, 3)
    kh = k.double().view(B, Tk, H, dk).permute(0, 2, 1, 3).clone().requires_grad_(True)
    vh = v.double().view(B, Tk, H, dk).permute(0, 2, 1, 3).clone().requires_grad_(True)
    if rel:
        qu = (qh + u.double()[None, :, None, :]).to(bf).double().requires_grad_(True)
        qv = (qh + vb.double()[None, :, None, :]).to(bf).double()
        ph = pos.double().view(-1, H, dk).permute(1, 2, 0)
        bd_full = (qv @ ph[None]).requires_grad_(True)
        idx = (Tq - 1) - torch.arange(Tq)[:, None] + torch.arange(Tk)[None, :]
        s = (qu @ kh.transpose(-1, -2) + torch.gather(bd_full, 3, idx[None, None].expand(B, H, Tq, Tk))) * scale
        qleaf = qu
    else:
        qleaf = qh.clone().requires_grad_(True)
        s = (qleaf @ kh.transpose(-1, -2)) * scale
    mask = torch.arange(Tk)[None, :] >= klen.long()[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    if causal:
        s = s + torch.triu(torch.full((Tq, Tk), float("-inf"), dtype=torch.float64), 1)
    pr = torch.softmax(s, -1)
    if pdrop > 0:
        keep = _dropout_mask(Z, Tq, Tk, drop).view(B, H, Tq, Tk)
        pr = pr * keep / (1.0 - pdrop)
    oref = pr @ vh
    (oref * dO.double().view(B, Tq, H, dk).permute(0, 2, 1, 3)).sum().backward()

    def rel_err(got, ref):
        return float((got - ref).norm() / ref.norm().clamp_min(1e-30))

    np.testing.assert_allclose(o.cpu().double().view(B, Tq, H, dk).permute(0, 2, 1, 3).numpy(), oref.detach().numpy(),
                               rtol=3e-2, atol=3e-2)
    got_dq = dq.cpu().double().view(B, Tq, H, dk).permute(0, 2, 1, 3)
    got_dk = dkk.cpu().double().view(B, Tk, H, dk).permute(0, 2, 1, 3)
    got_dv = dv.cpu().double().view(B, Tk, H, dk).permute(0, 2, 1, 3)
    assert rel_err(got_dq, qleaf.grad) < 1.5e-2
    assert rel_err(got_dk, kh.grad) < 1.5e-2
    assert rel_err(got_dv, vh.grad) < 1.5e-2
    # padded keys get exactly zero gradient
    for b in range(B):
        assert float(got_dk[b, :, int(klen[b]):].abs().max() if int(klen[b]) < Tk else 0.0) == 0.0
    if rel:
        got = dbd.cpu().double()[..., :2 * Tq - 1].permute(1, 0, 2, 3)  # (B, H, Tq, 2T-1)
        assert rel_err(got, bd_full.grad) < 1.5e-2
        assert float(dbd.cpu().double()[..., 2 * Tq - 1:].abs().max() if ldb > 2 * Tq - 1 else 0.0) == 0.0
        # ---- pos_pt: the dQ kernel adds the (Q+v) P^T branch itself and accumulates the two bias gradients
        OFF = 16
        ld_t = (OFF + 2 * Tq - 1 + 96 + 7) // 8 * 8
        ptb = torch.zeros(d, ld_t, dtype=bf, device=DEV)
        ptb[:, OFF:OFF + 2 * Tq - 1] = pp.t()
        du = torch.full((d,), 0.5, dtype=torch.float32, device=DEV)
        dvv = torch.full((d,), -0.25, dtype=torch.float32, device=DEV)
        dq3 = torch.full((B, Tq, d), 7.0, dtype=bf, device=DEV)
        K.attn_fused_bwd(qd, Tq * d, d, kd, Tk * d, d, vd, Tk * d, d, o, dOd, Tq * d, d, lse, torch.empty_like(delta), dq3,
                         torch.empty_like(dkk), torch.empty_like(dv), None, ldb, B, H, Tq, Tk, dk, kl, causal, scale, pp, d, ud,
                         vbd, drop, pos_pt=ptb[:, OFF:], pt_ld=ld_t, dpos_u=du, dpos_v=dvv)
        torch.cuda.synchronize()
        # gradient of the (Q+v) branch: dBD (reference) times the projected positions
        dqv_ref = bd_full.grad @ ph[None].transpose(-1, -2)          # (B, H, Tq, dk)
        tot_ref = qleaf.grad + dqv_ref
        got3 = dq3.cpu().double().view(B, Tq, H, dk).permute(0, 2, 1, 3)
        assert rel_err(got3, tot_ref) < 1.5e-2
        got_u = du.cpu().double().view(H, dk) - 0.5
        got_v = dvv.cpu().double().view(H, dk) + 0.25
        ref_u, ref_v = qleaf.grad.sum((0, 2)), dqv_ref.sum((0, 2))
        assert (got_u - ref_u).abs().max() <= 2e-2 * ref_u.abs().max() + 1e-4
        assert (got_v - ref_v).abs().max() <= 2e-2 * ref_v.abs().max() + 1e-4


@pytest.mark.parametrize("Tq,B,H", [(250, 3, 4), (17, 2, 2), (100, 2, 4), (251, 1, 4), (1004, 1, 2)])
def test_relpos_dqv(Tq, B, H):
    """s2t_relpos_dqv against float64: dqv from the band of the skewed dS, the in-place add into (a strided) dq, both bias
    gradients accumulated on top of what is there."""
    g = torch.Generator().manual_seed(Tq + H)
    dk, bf = 64, torch.bfloat16
    d = H * dk
    n_pos = 2 * Tq - 1
    ldb = (n_pos + 7) // 8 * 8
    # band-only content: row i holds columns Tq-1-i .. 2Tq-2-i, zero elsewhere (what the dQ kernel leaves)
    dbd = torch.zeros(H, B, Tq, ldb)
    ii = torch.arange(Tq)[:, None]
    nn = torch.arange(ldb)[None, :]
    band = (nn >= Tq - 1 - ii) & (nn <= 2 * Tq - 2 - ii)
    dbd = torch.where(band[None, None], torch.randn(H, B, Tq, ldb, generator=g) * 0.5, dbd).to(bf)
    p = (torch.randn(n_pos, d, generator=g) * 0.7).to(bf)
    off = 16
    pt_ld = (off + n_pos + 96 + 7) // 8 * 8
    pt = torch.zeros(d, pt_ld, dtype=bf)
    pt[:, off:off + n_pos] = p.t()
    ldq = 3 * d
    dqkv = (torch.randn(B * Tq, ldq, generator=g) * 0.5).to(bf)
    du0, dv0 = torch.randn(d, generator=g), torch.randn(d, generator=g)
    dq_dev, du, dv = dqkv.to(DEV), du0.to(DEV), dv0.to(DEV)
    pt_dev = pt.to(DEV)
    K.relpos_dqv(dbd.to(DEV), ldb, pt_dev[:, off:], pt_ld, dq_dev, Tq * ldq, ldq, du, dv, B, H, Tq, dk)
    torch.cuda.synchronize()
    # reference
    dqv = torch.einsum("hbin,nhc->bihc", dbd.double()[..., :n_pos], p.double().view(n_pos, H, dk)).reshape(B * Tq, d)
    old = dqkv.double()[:, :d]
    got = dq_dev.cpu().double()
    np.testing.assert_allclose(got[:, :d].numpy(), (old + dqv).numpy(), rtol=1e-2, atol=2e-2)  # one bf16 rounding of the sum
    np.testing.assert_array_equal(got[:, d:].numpy(), dqkv.double()[:, d:].numpy())  # k | v columns untouched
    np.testing.assert_allclose(du.cpu().double().numpy(), (du0.double() + old.sum(0)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(dv.cpu().double().numpy(), (dv0.double() + dqv.sum(0)).numpy(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("Tq,B,H", [(250, 3, 4), (17, 2, 2), (100, 2, 4), (251, 1, 4), (256, 2, 4),
                                    (257, 2, 2), (502, 2, 4), (1004, 1, 4), (700, 2, 1)])  # > 256: position rows in chunks of 512
def test_relpos_glue(Tq, B, H):
    """s2t_relpos_glue against float64: everything behind the skewed score gradient in one pass over it — dqv from the band of
    dbd added into a strided dq, both bias-gradient column sums accumulated on top of what is there (replicated workspace), and
    the gradient w.r.t. the projected positions dp[n][h, c] = sum_{b,i} dbd[h,b,i,n] qv[b,i,h,c] (overwritten; bf16 per-utterance
    partials summed in fp32)."""
    g = torch.Generator().manual_seed(3 * Tq + H)
    dk, bf = 64, torch.bfloat16
    d = H * dk
    n_pos = 2 * Tq - 1
    ldb = (n_pos + 7) // 8 * 8
    ii = torch.arange(Tq)[:, None]
    nn = torch.arange(ldb)[None, :]
    band = (nn >= Tq - 1 - ii) & (nn <= 2 * Tq - 2 - ii)
    dbd = torch.where(band[None, None], torch.randn(H, B, Tq, ldb, generator=g) * 0.5, torch.zeros(H, B, Tq, ldb)).to(bf)
    p = (torch.randn(n_pos, d, generator=g) * 0.7).to(bf)
    qv = (torch.randn(B * Tq, d, generator=g) * 0.6).to(bf)
    ldq = 3 * d
    dqkv = (torch.randn(B * Tq, ldq, generator=g) * 0.5).to(bf)
    R = 4
    ws0 = torch.randn(R, 2, d, generator=g)
    ws = ws0.clone().to(DEV)
    dq_dev = dqkv.to(DEV)
    dp = torch.full((n_pos, d), 7.0, device=DEV)
    K.relpos_glue(dbd.to(DEV), ldb, p.to(DEV), d, qv.to(DEV), dq_dev, Tq * ldq, ldq, ws.view(-1), ws.view(-1)[d:], dp, B, H, Tq, dk,
                  replicas=R, replica_stride=2 * d)
    torch.cuda.synchronize()
    dqv = torch.einsum("hbin,nhc->bihc", dbd.double()[..., :n_pos], p.double().view(n_pos, H, dk)).reshape(B * Tq, d)
    old = dqkv.double()[:, :d]
    got = dq_dev.cpu().double()
    # ONE bf16 rounding of the sum at any length: beyond 256 frames dq takes the share of every 512-row chunk of positions its
    # band meets in turn, and the remainder of the running bf16 sum travels with it (dq_lo)
    np.testing.assert_allclose(got[:, :d].numpy(), (old + dqv).numpy(), rtol=1e-2, atol=2e-2)
    np.testing.assert_array_equal(got[:, d:].numpy(), dqkv.double()[:, d:].numpy())  # k | v columns untouched
    add = (ws.cpu().double() - ws0.double()).sum(0)  # the column sums, whichever replica took them
    np.testing.assert_allclose(add[0].numpy(), old.sum(0).numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(add[1].numpy(), dqv.sum(0).numpy(), rtol=1e-4, atol=3e-3)
    dp_ref = torch.einsum("hbin,bihc->nhc", dbd.double()[..., :n_pos], qv.double().view(B, Tq, H, dk)).reshape(n_pos, d)
    err = (dp.cpu().double() - dp_ref).abs().max() / dp_ref.abs().max()
    assert err < 1e-2, float(err)  # bf16 partial per utterance
    # deferred reduce: two calls leave their partial tables, one s2t_relpos_dp_reduce launch sums both — the same bits
    dps = [torch.full((n_pos, d), 7.0, device=DEV), torch.full((n_pos, d), -3.0, device=DEV)]
    parts = [K.relpos_glue(dbd.to(DEV), ldb, p.to(DEV), d, qv.to(DEV), dqkv.to(DEV), Tq * ldq, ldq, ws.view(-1), ws.view(-1)[d:],
                           dps[k], B, H, Tq, dk, replicas=R, replica_stride=2 * d, defer_slot=k) for k in range(2)]
    assert float(dps[0].min()) == 7.0  # untouched until the reduce
    K.relpos_dp_reduce(parts, dps, B, H, Tq, dk)
    torch.cuda.synchronize()
    assert torch.equal(dps[0], dp) and torch.equal(dps[1], dp)


def test_relpos_glue_dp_precision_at_the_bench_batch():
    """The per-utterance partial tables of the position-table gradient leave the kernel rounded to bf16 and are summed in fp32
    (ADVICE round 3): at the bench shape (64 utterances of 250 frames, 4 heads) the sum of 64 independently rounded partials
    stays within 2e-3 relative L2 (5e-3 at the worst entry against the table's largest) of the float64 product."""
    Tq, B, H = 250, 64, 4
    g = torch.Generator().manual_seed(91)
    dk, bf = 64, torch.bfloat16
    d = H * dk
    n_pos = 2 * Tq - 1
    ldb = (n_pos + 7) // 8 * 8
    ii = torch.arange(Tq)[:, None]
    nn = torch.arange(ldb)[None, :]
    band = (nn >= Tq - 1 - ii) & (nn <= 2 * Tq - 2 - ii)
    dbd = torch.where(band[None, None], torch.randn(H, B, Tq, ldb, generator=g) * 0.5, torch.zeros(H, B, Tq, ldb)).to(bf)
    p = (torch.randn(n_pos, d, generator=g) * 0.7).to(bf)
    qv = (torch.randn(B * Tq, d, generator=g) * 0.6).to(bf)
    dq = torch.zeros(B * Tq, 3 * d, dtype=bf, device=DEV)
    ws = torch.zeros(4, 2, d, device=DEV)
    dp = torch.empty(n_pos, d, device=DEV)
    K.relpos_glue(dbd.to(DEV), ldb, p.to(DEV), d, qv.to(DEV), dq, Tq * 3 * d, 3 * d, ws.view(-1), ws.view(-1)[d:], dp, B, H, Tq, dk,
                  replicas=4, replica_stride=2 * d)
    torch.cuda.synchronize()
    ref = torch.einsum("hbin,bihc->nhc", dbd.to(DEV).double()[..., :n_pos], qv.to(DEV).double().view(B, Tq, H, dk)).reshape(n_pos, d)
    err = dp.double() - ref
    assert float(err.norm() / ref.norm()) < 2e-3
    assert float(err.abs().max() / ref.abs().max()) < 5e-3


@pytest.mark.parametrize("T,lens", [(250, [250, 201, 131, 64]), (502, [502, 430, 257, 90]), (1004, [1004, 640, 33])])
def test_relpos_glue_packed_rows_ignore_stale_columns(T, lens):
    """Packed batch (include/s2t_hip.h "Packed rows"): dq rows of utterance b from cu[b]; dbd keeps its padded [H][B][T] slab,
    of which this pass's dQ kernel wrote rows i < cap_b and columns n < T-1-i+cap_b only — everything else may be the band of a
    longer utterance of an earlier batch and must not be read.  The slab is filled with such garbage here."""
    from s2t_amd import rows as Rows

    H, halo = 4, 7
    B = len(lens)
    cap = [min(l + halo, T) for l in lens]
    cu = [0]
    for c in cap:
        cu.append(cu[-1] + c)
    g = torch.Generator().manual_seed(17)
    dk, bf = 64, torch.bfloat16
    d = H * dk
    n_pos = 2 * T - 1
    ldb = (n_pos + 7) // 8 * 8
    ii = torch.arange(T)[None, :, None]
    nn = torch.arange(ldb)[None, None, :]
    capt = torch.tensor(cap)[:, None, None]
    written = (ii < capt) & (nn >= T - 1 - ii) & (nn < T - 1 - ii + capt)                    # what this pass's dQ kernel wrote
    band = (nn >= T - 1 - ii) & (nn <= 2 * T - 2 - ii)                                        # where stale values may sit
    vals = torch.randn(H, B, T, ldb, generator=g) * 0.5
    junk = torch.randn(H, B, T, ldb, generator=g) * 3.0
    dbd = torch.where(written[None], vals, torch.where(band[None].expand(1, B, T, ldb), junk, torch.zeros(()))).to(bf)
    clean = torch.where(written[None], vals, torch.zeros(())).to(bf)
    p = (torch.randn(n_pos, d, generator=g) * 0.7).to(bf)
    qv = (torch.randn(B * T, d, generator=g) * 0.6).to(bf)                                     # padded row order (b * T + i)
    ldq = 3 * d
    dqkv = (torch.randn(B * T, ldq, generator=g) * 0.5).to(bf)                                 # packed rows: utterance b at cu[b]
    lens32 = torch.tensor(lens, dtype=torch.int32, device=DEV)
    Rows.attach(lens32, B, T, halo)
    assert lens32._pk.cu.tolist() == cu
    ws = torch.zeros(4, 2, d, device=DEV)
    dq_dev = dqkv.to(DEV)
    dp = torch.empty(n_pos, d, device=DEV)
    K.relpos_glue(dbd.to(DEV), ldb, p.to(DEV), d, qv.to(DEV), dq_dev, T * ldq, ldq, ws.view(-1), ws.view(-1)[d:], dp, B, H, T, dk,
                  replicas=4, replica_stride=2 * d, rows=lens32)
    torch.cuda.synchronize()
    dqv = torch.einsum("hbin,nhc->bihc", clean.double()[..., :n_pos], p.double().view(n_pos, H, dk))  # [B, T, H, dk]
    got = dq_dev.cpu().double()
    want = dqkv.double().clone()
    for b in range(B):
        want[cu[b]:cu[b] + cap[b], :d] += dqv[b, :cap[b]].reshape(cap[b], d)
    np.testing.assert_allclose(got[:cu[-1], :d].numpy(), want[:cu[-1], :d].numpy(), rtol=1e-2, atol=2e-2)
    np.testing.assert_array_equal(got[cu[-1]:].numpy(), dqkv.double()[cu[-1]:].numpy())       # rows beyond the live ones untouched
    np.testing.assert_array_equal(got[:, d:].numpy(), dqkv.double()[:, d:].numpy())
    dp_ref = torch.einsum("hbin,bihc->nhc", clean.double()[..., :n_pos], qv.double().view(B, T, H, dk)).reshape(n_pos, d)
    err = (dp.cpu().double() - dp_ref).abs().max() / dp_ref.abs().max()
    assert err < 1e-2, float(err)


@pytest.mark.parametrize("kind", ["self_rel", "self_causal", "cross_packed_keys", "cross_packed_both"])
@pytest.mark.parametrize("pdrop", [0.0, 0.1])
def test_fused_attention_on_packed_rows_equals_the_padded_layout(kind, pdrop):
    """s2t_attn_fused_fwd / _bwd with ``cu_q`` / ``cu_k`` (include/s2t_hip.h "Packed rows"): utterance b's queries / keys start at
    row cu[b] of a packed matrix instead of b * T.  The same values through both layouts give the same bits on every row that
    holds a frame (and the same dropout mask: lse, delta and the mask index keep the padded strides): o, dq, dk, dv, the skewed
    score gradient inside the written band and the rounded Q + pos_bias_v rows."""
    from s2t_amd import rows as Rows

    g = torch.Generator().manual_seed(len(kind) + int(pdrop * 100))
    B, H, dk, bf = 5, 4, 64, torch.bfloat16
    d = H * dk
    rel, causal = kind == "self_rel", kind == "self_causal"
    Tq = 61 if kind.startswith("cross") or causal else 250
    Tk = Tq if kind.startswith("self") else 250
    qlens = [61, 50, 33, 12, 1] if Tq == 61 else [250, 243, 180, 65, 9]
    klens = qlens if kind.startswith("self") else [250, 243, 180, 65, 9]
    halo = 7 if rel else 0
    pack_q = kind != "cross_packed_keys"
    ql = torch.tensor(qlens, dtype=torch.int32, device=DEV)
    kl = ql if kind.startswith("self") else torch.tensor(klens, dtype=torch.int32, device=DEV)
    if pack_q:
        Rows.attach(ql, B, Tq, halo)
    if kl is not ql:
        Rows.attach(kl, B, Tk, 0)
    q = (torch.randn(B * Tq, d, generator=g) * 0.7).to(bf).to(DEV)
    k = (torch.randn(B * Tk, d, generator=g) * 0.7).to(bf).to(DEV)
    v = (torch.randn(B * Tk, d, generator=g) * 0.7).to(bf).to(DEV)
    dO = (torch.randn(B * Tq, d, generator=g) * 0.5).to(bf).to(DEV)
    qmask = (torch.arange(Tq, device=DEV)[None, :] < ql[:, None]).reshape(-1)
    kmask = (torch.arange(Tk, device=DEV)[None, :] < kl[:, None]).reshape(-1)
    dO[~qmask] = 0  # padded queries carry no gradient (what the model's masks make of them)
    pos = u = vb = None
    if rel:
        pos = (torch.randn(2 * Tq - 1, d, generator=g) * 0.7).to(bf).to(DEV)
        u = (torch.randn(d, generator=g) * 0.3).to(DEV)
        vb = (torch.randn(d, generator=g) * 0.3).to(DEV)
    seed = torch.full((1,), 77, dtype=torch.int64, device=DEV)
    drop = (pdrop, seed, 3) if pdrop > 0 else None
    scale = 1.0 / math.sqrt(dk)
    Z = B * H
    ldb = (2 * Tq - 1 + 7) // 8 * 8

    def run(packed):
        qq = Rows.pack(q, ql) if (packed and pack_q) else q
        kk = Rows.pack(k, kl) if packed else k
        vv = Rows.pack(v, kl) if packed else v
        dd = Rows.pack(dO, ql) if (packed and pack_q) else dO
        qr = ql if (packed and pack_q) else None
        kr = kl if packed else None
        key_lens = kl if packed else Rows.detached(kl)
        o = torch.zeros(B * Tq, d, dtype=bf, device=DEV)
        lse = torch.zeros(Z, Tq, dtype=torch.float32, device=DEV)
        K.attn_fused_fwd(qq, Tq * d, d, kk, Tk * d, d, vv, Tk * d, d, o, Tq * d, d, lse, B, H, Tq, Tk, dk, key_lens, causal, scale,
                         pos, d if rel else 0, u, vb, drop, q_rows=qr, k_rows=kr)
        delta = torch.zeros(Z, Tq, dtype=torch.float32, device=DEV)
        dq = torch.zeros(B * Tq, d, dtype=bf, device=DEV)
        dk_ = torch.zeros(B * Tk, d, dtype=bf, device=DEV)
        dv = torch.zeros(B * Tk, d, dtype=bf, device=DEV)
        dbd = torch.zeros(H, B, Tq, ldb, dtype=bf, device=DEV) if rel else None
        qv = torch.zeros(B * Tq, d, dtype=bf, device=DEV) if rel else None
        K.attn_fused_bwd(qq, Tq * d, d, kk, Tk * d, d, vv, Tk * d, d, o, dd, Tq * d, d, lse, delta, dq, dk_, dv, dbd, ldb, B, H,
                         Tq, Tk, dk, key_lens, causal, scale, pos, d if rel else 0, u, vb, drop, dbd_band_only=rel, qv_out=qv,
                         q_rows=qr, k_rows=kr)
        torch.cuda.synchronize()
        if packed:
            if pack_q:
                o, dq = Rows.unpack(o, ql), Rows.unpack(dq, ql)
            dk_, dv = Rows.unpack(dk_, kl), Rows.unpack(dv, kl)
        return o, dq, dk_, dv, dbd, qv, lse

    a, b = run(False), run(True)
    assert torch.equal(a[0][qmask], b[0][qmask]), "o"
    assert torch.equal(a[1][qmask], b[1][qmask]), "dq"
    assert torch.equal(a[2][kmask], b[2][kmask]), "dk"
    assert torch.equal(a[3][kmask], b[3][kmask]), "dv"
    lmask = qmask.view(B, Tq)[:, None, :].expand(B, H, Tq).reshape(Z, Tq)
    assert torch.equal(a[6][lmask], b[6][lmask]), "lse"
    assert float(b[0][qmask].float().abs().max()) > 0 and float(b[2][kmask].float().abs().max()) > 0
    if rel:
        assert torch.equal(a[5][qmask], b[5][qmask]), "q + pos_bias_v"
        # the band of row i the packed pass wrote: columns T-1-i .. T-1-i+cap_b-1; inside it the two layouts agree
        cap = torch.tensor([min(l + halo, Tq) for l in qlens], device=DEV)
        ii = torch.arange(Tq, device=DEV)[None, :, None]
        nn_ = torch.arange(ldb, device=DEV)[None, None, :]
        written = (ii < cap[:, None, None]) & (nn_ >= Tq - 1 - ii) & (nn_ < Tq - 1 - ii + cap[:, None, None])
        w4 = written[None].expand(H, B, Tq, ldb)
        assert torch.equal(a[4][w4], b[4][w4]), "dbd"


def test_strides_beyond_the_24_bit_products_are_refused():
    """The tile loads form row x stride with 24-bit multiplies: a row stride (or a length) of 65 536 elements or more is refused
    with S2T_ERR_UNSUPPORTED instead of reading the wrong rows."""
    B, H, T, dk = 1, 1, 16, 64
    q = torch.zeros(T, 65536 + 64, dtype=torch.bfloat16, device=DEV)
    o = torch.empty(T, 64, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(T, device=DEV)
    with pytest.raises(RuntimeError, match="s2t_attn_fused_fwd"):
        K.attn_fused_fwd(q, T * q.shape[1], q.shape[1], q, T * q.shape[1], q.shape[1], q, T * q.shape[1], q.shape[1], o, T * 64, 64,
                         lse, B, H, T, T, dk, None, False, 0.125)
    qs = q[:, :64].contiguous()
    K.attn_fused_fwd(qs, T * 64, 64, qs, T * 64, 64, qs, T * 64, 64, o, T * 64, 64, lse, B, H, T, T, dk, None, False, 0.125)
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()


def test_position_offsets_beyond_32_bits_are_refused():
    """Relative positions: row n of the projected table runs to 2 Tq - 2 and its offset n * p_sr is a 24-bit product kept in 32
    bits; a (length, stride) pair whose product leaves 32 bits is refused instead of wrapping."""
    B, H, T, dk = 1, 1, 40000, 64
    t = torch.zeros(64, 64, dtype=torch.bfloat16, device=DEV)  # never read: the call must fail on its arguments
    lse = torch.empty(64, device=DEV)
    u = torch.zeros(64, device=DEV)
    with pytest.raises(RuntimeError, match="s2t_attn_fused_fwd"):
        K.attn_fused_fwd(t, T * 64, 64, t, T * 64, 64, t, T * 64, 64, t, T * 64, 64, lse, B, H, T, T, dk, None, False, 0.125,
                         t, 65528, u, u, None)


def test_an_utterance_without_rows_in_a_packed_batch():
    """cu[b] == cu[b + 1]: the utterance's workgroups read and store nothing (the clamped tile rows must not go to -1), the
    other utterances' outputs are those of the same batch without it.  Plain and relative-position self-attention forward."""
    from s2t_amd import rows as Rows

    H, dk, T = 4, 64, 96
    d = H * dk
    live = [96, 50]

    def run(lens, rel):
        B = len(lens)
        lt = torch.tensor(lens, dtype=torch.int32, device=DEV)
        Rows.attach(lt, B, T, 0)
        M = B * T
        g = torch.Generator(device="cpu").manual_seed(3)
        qkv = torch.randn(sum(live), 3 * d, generator=g).bfloat16()
        buf = torch.zeros(M, 3 * d, dtype=torch.bfloat16)
        buf[:sum(live)] = qkv
        buf = buf.to(DEV)
        O = torch.full((M, d), 7.0, dtype=torch.bfloat16, device=DEV)
        lse = torch.zeros(B * H, T, dtype=torch.float32, device=DEV)
        pos = u = v = None
        if rel:
            pos = torch.randn(2 * T - 1, d, generator=g).bfloat16().to(DEV)
            u = torch.randn(d, generator=g).to(DEV)
            v = torch.randn(d, generator=g).to(DEV)
        K.attn_fused_fwd(buf, T * 3 * d, 3 * d, buf[:, d:], T * 3 * d, 3 * d, buf[:, 2 * d:], T * 3 * d, 3 * d, O, T * d, d, lse,
                         B, H, T, T, dk, lt, False, 0.125, pos, d if rel else 0, u, v, None, q_rows=lt, k_rows=lt)
        torch.cuda.synchronize()
        return O[:sum(live)].clone(), O[sum(live):]

    for rel in (False, True):
        oa, rest_a = run([96, 0, 50], rel)
        ob, _ = run([96, 50], rel)
        assert torch.equal(oa, ob), rel
        assert bool((rest_a.float() == 7.0).all()), rel  # nothing stored beyond the live rows


@pytest.mark.parametrize("T,B,H,pdrop,lens", [(250, 3, 4, 0.0, None), (250, 3, 4, 0.1, None), (256, 2, 4, 0.1, None),
                                              (100, 2, 2, 0.0, None), (17, 2, 2, 0.1, None), (33, 1, 4, 0.0, None),
                                              (251, 2, 4, 0.1, None), (3, 2, 2, 0.0, None), (64, 2, 8, 0.1, None),
                                              (250, 4, 4, 0.1, [250, 201, 131, 30]), (256, 3, 2, 0.0, [256, 97, 1]),
                                              (251, 3, 2, 0.1, [251, 64, 33])])
def test_relpos_backward_in_one_pass(T, B, H, pdrop, lens):
    """s2t_relpos_attn_bwd (csrc/relpos_bwd.hip; espnet_multihead_attention.py:292-356 backward): dq (both branches), dk, dv,
    the two position-bias gradients and the gradient w.r.t. the projected positions from ONE launch, against float64 autograd on
    the bf16-rounded operands — uniform rows with padded keys (lens = None: key_lens masks) and a packed batch (lens: rows of
    utterance b from cu[b]; stale values outside the utterances must not be read or written)."""
    g = torch.Generator().manual_seed(T * 7 + B + int(pdrop * 100))
    dk = 64
    d = H * dk
    Z = B * H
    bf = torch.bfloat16
    packed = lens is not None
    klen = torch.tensor(lens if packed else [T, max(1, T - 7), max(1, T // 2), 5][:B], dtype=torch.int32)
    q = (torch.randn(B, T, d, generator=g) * 0.7).to(bf)
    k = (torch.randn(B, T, d, generator=g) * 0.7).to(bf)
    v = (torch.randn(B, T, d, generator=g) * 0.7).to(bf)
    dO = (torch.randn(B, T, d, generator=g) * 0.5).to(bf)
    pos = (torch.randn(2 * T - 1, d, generator=g) * 0.7).to(bf)
    u = torch.randn(H, dk, generator=g) * 0.3
    vb = torch.randn(H, dk, generator=g) * 0.3
    scale = 1.0 / math.sqrt(dk)
    seed = torch.full((1,), 4321, dtype=torch.int64, device=DEV)
    drop = (pdrop, seed, 3) if pdrop > 0 else None
    if packed:
        for b in range(B):  # rows beyond an utterance carry no gradient and are not part of the problem
            dO[b, int(klen[b]):] = 0
    pp, ud, vbd, kl = pos.to(DEV), u.reshape(-1).to(DEV), vb.reshape(-1).to(DEV), klen.to(DEV)
    o = torch.empty(B, T, d, dtype=bf, device=DEV)
    lse = torch.empty(Z, T, dtype=torch.float32, device=DEV)
    qd, kd, vd, dOd = q.to(DEV), k.to(DEV), v.to(DEV), dO.to(DEV)
    K.attn_fused_fwd(qd, T * d, d, kd, T * d, d, vd, T * d, d, o, T * d, d, lse, B, H, T, T, dk, kl, False, scale, pp, d, ud, vbd,
                     drop)
    R = 4
    du = torch.full((R, 2, d), 0.5, dtype=torch.float32, device=DEV)

    class _Rows:  # what kernels._cu() reads
        pass

    if packed:
        cu = torch.zeros(B + 1, dtype=torch.int32)
        cu[1:] = torch.cumsum(klen, 0)
        n = int(cu[-1])

        def pack(t, fill=None):
            out = torch.full((n + 3, d), 7.0 if fill is None else fill, dtype=bf, device=DEV)
            for b in range(B):
                out[int(cu[b]):int(cu[b + 1])] = t[b, :int(klen[b])]
            return out

        qp, kp, vp, op, dop = pack(qd), pack(kd), pack(vd), pack(o), pack(dOd)
        dqp, dkp, dvp = pack(qd, 7.0), pack(qd, 7.0), pack(qd, 7.0)
        dqp.fill_(7.0); dkp.fill_(7.0); dvp.fill_(7.0)
        from s2t_amd import rows as Rows
        rows = Rows.attach(kl.clone(), B, T, 0)
        # (the geometry's cu must be this test's: no halo rows)
        assert torch.equal(K.rows_geom(rows).cu.cpu()[:B + 1], cu)
        part = K.relpos_attn_bwd(qp, 0, d, kp, 0, d, vp, 0, d, op, dop, 0, d, lse, dqp, dkp, dvp, pp, d, ud, vbd, du.view(-1),
                                 du.view(-1)[d:], B, H, T, dk, rows, scale, drop, replicas=R, replica_stride=2 * d, rows=rows)
        torch.cuda.synchronize()
        assert float(dqp[n:].float().min()) == 7.0 and float(dkp[n:].float().min()) == 7.0 and float(dvp[n:].float().min()) == 7.0

        def unpack(t):
            out = torch.zeros(B, T, d, dtype=bf, device=DEV)
            for b in range(B):
                out[b, :int(klen[b])] = t[int(cu[b]):int(cu[b + 1])]
            return out

        dq, dkk, dv = unpack(dqp), unpack(dkp), unpack(dvp)
    else:
        dq = torch.full((B, T, d), 7.0, dtype=bf, device=DEV)
        dkk = torch.full((B, T, d), 7.0, dtype=bf, device=DEV)
        dv = torch.full((B, T, d), 7.0, dtype=bf, device=DEV)
        part = K.relpos_attn_bwd(qd, T * d, d, kd, T * d, d, vd, T * d, d, o, dOd, T * d, d, lse, dq, dkk, dv, pp, d, ud, vbd,
                                 du.view(-1), du.view(-1)[d:], B, H, T, dk, kl, scale, drop, replicas=R, replica_stride=2 * d)
    dp = torch.full((2 * T - 1, d), 3.0, dtype=torch.float32, device=DEV)
    K.relpos_dp_reduce([part], [dp], B, H, T, dk)
    torch.cuda.synchronize()

    # ---- float64 autograd reference on the same bf16-rounded operands
    qh = q.double().view(B, T, H, dk).permute(0, 2, 1, 3)
    kh = k.double().view(B, T, H, dk).permute(0, 2, 1, 3).clone().requires_grad_(True)
    vh = v.double().view(B, T, H, dk).permute(0, 2, 1, 3).clone().requires_grad_(True)
    qu = (qh + u.double()[None, :, None, :]).to(bf).double().requires_grad_(True)
    qv = (qh + vb.double()[None, :, None, :]).to(bf).double().requires_grad_(True)
    ph = pos.double().view(-1, H, dk).permute(1, 2, 0).clone().requires_grad_(True)  # (H, dk, 2T-1)
    bd_full = qv @ ph[None]
    idx = (T - 1) - torch.arange(T)[:, None] + torch.arange(T)[None, :]
    s = (qu @ kh.transpose(-1, -2) + torch.gather(bd_full, 3, idx[None, None].expand(B, H, T, T))) * scale
    mask = torch.arange(T)[None, :] >= klen.long()[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    pr = torch.softmax(s, -1)
    if pdrop > 0:
        keep = _dropout_mask(Z, T, T, drop).view(B, H, T, T)
        pr = pr * keep / (1.0 - pdrop)
    oref = pr @ vh
    (oref * dO.double().view(B, T, H, dk).permute(0, 2, 1, 3)).sum().backward()

    def rel_err(got, ref):
        return float((got - ref).norm() / ref.norm().clamp_min(1e-30))

    valid = (torch.arange(T)[None, :] < klen.long()[:, None]) if packed else torch.ones(B, T, dtype=torch.bool)
    vm = valid[:, None, :, None].double()
    got_dq = dq.cpu().double().view(B, T, H, dk).permute(0, 2, 1, 3)
    got_dk = dkk.cpu().double().view(B, T, H, dk).permute(0, 2, 1, 3)
    got_dv = dv.cpu().double().view(B, T, H, dk).permute(0, 2, 1, 3)
    ref_dq = (qu.grad + qv.grad) * vm
    assert rel_err(got_dq * vm, ref_dq) < 1.5e-2
    assert rel_err(got_dk, kh.grad) < 1.5e-2
    assert rel_err(got_dv, vh.grad) < 1.5e-2
    for b in range(B):  # padded keys get exactly zero gradient
        if int(klen[b]) < T:
            assert float(got_dk[b, :, int(klen[b]):].abs().max()) == 0.0 and float(got_dv[b, :, int(klen[b]):].abs().max()) == 0.0
    got_u = du[:, 0].sum(0).cpu().double().view(H, dk) - 0.5 * R
    got_v = du[:, 1].sum(0).cpu().double().view(H, dk) - 0.5 * R
    ref_u, ref_v = (qu.grad * vm).sum((0, 2)), (qv.grad * vm).sum((0, 2))
    assert (got_u - ref_u).abs().max() <= 2e-2 * ref_u.abs().max() + 1e-4
    assert (got_v - ref_v).abs().max() <= 2e-2 * ref_v.abs().max() + 1e-4
    ref_dp = ph.grad.permute(2, 0, 1).reshape(2 * T - 1, d)  # (2T-1, H*dk)
    assert rel_err(dp.cpu().double(), ref_dp) < 1.5e-2


def test_relpos_backward_in_one_pass_with_an_empty_utterance_and_refused_shapes():
    """A packed batch may hold an utterance without rows (cu[b] == cu[b + 1]): nothing of it is read, its gradient rows do not
    exist, and its table of the position gradient is ZERO (the reduction sums every utterance's table) — the other utterances'
    results equal those of the batch without it.  Sequences beyond 256 frames are refused (S2T_ERR_UNSUPPORTED: the three-kernel
    route takes them)."""
    from s2t_amd import rows as Rows

    T, H, dk = 100, 2, 64
    d = H * dk
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(77)
    scale = 1.0 / math.sqrt(dk)
    pos = (torch.randn(2 * T - 1, d, generator=g) * 0.7).to(bf).to(DEV)
    u = (torch.randn(d, generator=g) * 0.3).to(DEV)
    vb = (torch.randn(d, generator=g) * 0.3).to(DEV)

    def run(lens):
        B = len(lens)
        n = sum(lens)
        gg = torch.Generator().manual_seed(5)
        rowsd = {k_: (torch.randn(sum(l for l in lens if l), d, generator=gg) * 0.6).to(bf).to(DEV) for k_ in "qkvoD"}
        lse = torch.zeros(B * H, T, dtype=torch.float32, device=DEV)
        lse_live = torch.randn(sum(1 for l in lens if l) * H, T, generator=gg).to(DEV) + 6.0
        j = 0
        for b, l in enumerate(lens):
            if l:
                lse[b * H:(b + 1) * H] = lse_live[j * H:(j + 1) * H]
                j += 1
        kl = torch.tensor(lens, dtype=torch.int32, device=DEV)
        rows = Rows.attach(kl, B, T, 0)
        dq = torch.full((n + 2, d), 7.0, dtype=bf, device=DEV)
        dkk, dv = dq.clone(), dq.clone()
        du = torch.zeros(2, d, dtype=torch.float32, device=DEV)
        part = K.relpos_attn_bwd(rowsd["q"], 0, d, rowsd["k"], 0, d, rowsd["v"], 0, d, rowsd["o"], rowsd["D"], 0, d, lse, dq, dkk, dv,
                                 pos, d, u, vb, du.view(-1), du.view(-1)[d:], B, H, T, dk, rows, scale, None, rows=rows)
        dp = torch.empty(2 * T - 1, d, dtype=torch.float32, device=DEV)
        K.relpos_dp_reduce([part], [dp], B, H, T, dk)
        torch.cuda.synchronize()
        tables = part.view(torch.bfloat16)[:B * (2 * T - 1) * d].view(B, 2 * T - 1, d).float().clone()
        return dq[:n].float(), dkk[:n].float(), dv[:n].float(), du.clone(), dp.clone(), tables, (dq[n:], dkk[n:], dv[n:])

    a = run([100, 0, 37])
    b = run([100, 37])
    for x_, y_ in zip(a[:3], b[:3]):
        assert torch.equal(x_, y_)
    assert torch.equal(a[3], b[3]) and torch.allclose(a[4], b[4], rtol=0, atol=0)
    assert float(a[5][1].abs().max()) == 0.0 and torch.equal(a[5][0], b[5][0]) and torch.equal(a[5][2], b[5][1])
    assert all(float(t.float().min()) == 7.0 and float(t.float().max()) == 7.0 for t in a[6])
    # beyond 256 frames: refused
    Tl = 257
    z = torch.zeros(Tl, d, dtype=bf, device=DEV)
    with pytest.raises(RuntimeError):
        K.relpos_attn_bwd(z, Tl * d, d, z, Tl * d, d, z, Tl * d, d, z, z, Tl * d, d, torch.zeros(H, Tl, device=DEV), z.clone(), z.clone(),
                          z.clone(), torch.zeros(2 * Tl - 1, d, dtype=bf, device=DEV), d, u, vb, torch.zeros(d, device=DEV),
                          torch.zeros(d, device=DEV), 1, H, Tl, dk, None, scale, None)


@pytest.mark.parametrize("Tq,Tk,causal,pdrop,use_lo", [(250, 250, False, 0.0, False), (61, 61, True, 0.0, False),
                                                       (61, 250, False, 0.0, True), (130, 130, False, 0.1, False),
                                                       (91, 249, False, 0.1, True), (33, 33, True, 0.1, False),
                                                       (300, 256, False, 0.0, False)])
def test_plain_backward_in_one_pass(Tq, Tk, causal, pdrop, use_lo):
    """s2t_attn_bwd_one_pass (csrc/relpos_bwd.hip without its position terms; multihead_attention.py:161-431 backward): dq, dk, dv
    of plain attention from ONE launch against float64 autograd on the bf16-rounded operands — self-attention, the decoder's causal
    form, encoder-decoder shapes (Tq != Tk, odd Tk: the general dropout hash), with the forward's rounding remainder o_lo — and
    against the two kernels of s2t_attn_fused_bwd (same inputs: within the two routes' roundings)."""
    g = torch.Generator().manual_seed(Tq * 13 + Tk + int(pdrop * 10))
    B, H, dk = 3, 4, 64
    d = H * dk
    Z = B * H
    bf = torch.bfloat16
    q = (torch.randn(B, Tq, d, generator=g) * 0.7).to(bf)
    k = (torch.randn(B, Tk, d, generator=g) * 0.7).to(bf)
    v = (torch.randn(B, Tk, d, generator=g) * 0.7).to(bf)
    dO = (torch.randn(B, Tq, d, generator=g) * 0.5).to(bf)
    klen = torch.tensor([Tk, max(1, Tk - 7), max(1, Tk // 2)], dtype=torch.int32)
    scale = dk ** -0.5
    seed = torch.full((1,), 99, dtype=torch.int64, device=DEV)
    drop = (pdrop, seed, 4) if pdrop > 0 else None
    qd, kd, vd, dOd, kl = q.to(DEV), k.to(DEV), v.to(DEV), dO.to(DEV), klen.to(DEV)
    o = torch.empty(B, Tq, d, dtype=bf, device=DEV)
    o_lo = torch.empty_like(o) if use_lo else None
    lse = torch.empty(Z, Tq, dtype=torch.float32, device=DEV)
    K.attn_fused_fwd(qd, Tq * d, d, kd, Tk * d, d, vd, Tk * d, d, o, Tq * d, d, lse, B, H, Tq, Tk, dk, kl, causal, scale, None, 0,
                     None, None, drop, o_lo=o_lo)
    dq = torch.full((B, Tq, d), 7.0, dtype=bf, device=DEV)
    dkk = torch.full((B, Tk, d), 7.0, dtype=bf, device=DEV)
    dv = torch.full((B, Tk, d), 7.0, dtype=bf, device=DEV)
    K.attn_bwd_one_pass(qd, Tq * d, d, kd, Tk * d, d, vd, Tk * d, d, o, dOd, Tq * d, d, lse, dq, dkk, dv, B, H, Tq, Tk, dk, kl, causal,
                        scale, drop, o_lo=o_lo)
    # the two-kernel route on the same inputs
    delta = torch.empty(Z, Tq, dtype=torch.float32, device=DEV)
    dq2, dk2, dv2 = torch.empty_like(dq), torch.empty_like(dkk), torch.empty_like(dv)
    K.attn_fused_bwd(qd, Tq * d, d, kd, Tk * d, d, vd, Tk * d, d, o, dOd, Tq * d, d, lse, delta, dq2, dk2, dv2, None, 0, B, H, Tq, Tk,
                     dk, kl, causal, scale, None, 0, None, None, drop, o_lo=o_lo)
    torch.cuda.synchronize()

    qh = q.double().view(B, Tq, H, dk).permute(0, 2, 1, 3).clone().requires_grad_(True)
    kh = k.double().view(B, Tk, H, dk).permute(0, 2, 1, 3).clone().requires_grad_(True)
    vh = v.double().view(B, Tk, H, dk).permute(0, 2, 1, 3).clone().requires_grad_(True)
    s = (qh @ kh.transpose(-1, -2)) * scale
    s = s.masked_fill((torch.arange(Tk)[None, :] >= klen.long()[:, None])[:, None, None, :], float("-inf"))
    if causal:
        s = s + torch.triu(torch.full((Tq, Tk), float("-inf"), dtype=torch.float64), 1)
    pr = torch.softmax(s, -1)
    if pdrop > 0:
        pr = pr * _dropout_mask(Z, Tq, Tk, drop).view(B, H, Tq, Tk) / (1.0 - pdrop)
    ((pr @ vh) * dO.double().view(B, Tq, H, dk).permute(0, 2, 1, 3)).sum().backward()

    def rel_err(got, ref):
        return float((got - ref).norm() / ref.norm().clamp_min(1e-30))

    hq = lambda t, T_: t.cpu().double().view(B, T_, H, dk).permute(0, 2, 1, 3)
    e1 = (rel_err(hq(dq, Tq), qh.grad), rel_err(hq(dkk, Tk), kh.grad), rel_err(hq(dv, Tk), vh.grad))
    e2 = (rel_err(hq(dq2, Tq), qh.grad), rel_err(hq(dk2, Tk), kh.grad), rel_err(hq(dv2, Tk), vh.grad))
    print("one pass", ["%.2e" % e for e in e1], "two kernels", ["%.2e" % e for e in e2])
    assert max(e1) < 1.5e-2
    assert all(a_ <= 1.5 * b_ + 1e-3 for a_, b_ in zip(e1, e2))
    for b in range(B):  # padded keys get exactly zero gradient
        if int(klen[b]) < Tk:
            assert float(hq(dkk, Tk)[b, :, int(klen[b]):].abs().max()) == 0.0 and float(hq(dv, Tk)[b, :, int(klen[b]):].abs().max()) == 0.0
