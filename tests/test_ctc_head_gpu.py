"""s2t_ctc_head_greedy (csrc/ctc_head.hip): the CTC head and the greedy arg-max in one launch — per row the first arg-max of
x W^T + b, its log-probability and the logsumexp, the [rows, V] fp32 logits never stored
(modules/speech_to_text/ctc.py:60-63 + models/speech_to_text/s2t_ctc.py:312-328) — against float64 on the same bf16 operands and
against the two-kernel route it replaces (s2t_gemm with an fp32 output + s2t_argmax_lse); CTCDecoder.generate with the fused head
against the same decoder without it, at configuration 5a's model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from s2t_amd import kernels as K  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402

DEV = "cuda"


def _near_tie_free(ref64, idx_ref, got, tol):
    """rows whose ids differ must be exact ties to within ``tol`` of the logit scale in the float64 reference"""
    bad = (got != idx_ref).nonzero().flatten()
    for r in bad.tolist():
        row = ref64[r]
        assert float(row[idx_ref[r]] - row[got[r]]) <= tol * float(row.abs().max()), (r, int(idx_ref[r]), int(got[r]))
    return bad.numel()


@pytest.mark.parametrize("Mrows,V,bias", [(64, 128, True), (1000, 10000, True), (4097, 10000, False), (333, 257, True), (12950, 10000, True)])
def test_head_greedy_against_float64_and_the_two_kernel_route(Mrows, V, bias):
    g = torch.Generator().manual_seed(Mrows + V)
    x = torch.randn(Mrows, 256, generator=g).bfloat16()
    w = (torch.randn(V, 256, generator=g) * 256 ** -0.5).bfloat16()
    b = (torch.randn(V, generator=g) * 0.5) if bias else None
    # a planted exact tie: two vocabulary entries with identical weights and bias -> the LOWER index must win
    if V >= 300:
        w[V - 7] = w[11]
        if b is not None:
            b[V - 7] = b[11]
    xd, wd = x.to(DEV), w.to(DEV)
    bd = b.to(DEV) if b is not None else None
    idx = torch.full((Mrows,), -5, dtype=torch.int32, device=DEV)
    top = torch.full((Mrows,), 7.0, dtype=torch.float32, device=DEV)
    lse = torch.full((Mrows,), 7.0, dtype=torch.float32, device=DEV)
    K.ctc_head_greedy(xd, wd, bd, idx, top, lse)
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t()
    if b is not None:
        ref = ref + b.double()
    lse_ref = torch.logsumexp(ref, 1)
    mx_ref, idx_ref = ref.max(1)
    # torch.max returns an arbitrary index among exact ties on some back ends: take the FIRST maximum explicitly
    idx_ref = (ref == mx_ref[:, None]).float().argmax(1)
    got = idx.cpu().long()
    n_bad = _near_tie_free(ref, idx_ref, got, 1e-5)
    assert n_bad <= max(2, Mrows // 2000), n_bad
    np.testing.assert_allclose(lse.cpu().double().numpy(), lse_ref.numpy(), rtol=0, atol=2e-4)
    same = got == idx_ref
    np.testing.assert_allclose(top.cpu().double()[same].numpy(), (mx_ref - lse_ref)[same].numpy(), rtol=0, atol=3e-4)
    if V >= 300:  # rows where entry 11 (and its twin V - 7) is the maximum: the lower index
        hit = (idx_ref == 11).nonzero().flatten()
        assert all(int(got[r]) == 11 for r in hit.tolist())
    # the route it replaces: fp32 logits through s2t_gemm, then s2t_argmax_lse
    Vp = (V + 7) // 8 * 8
    logits = torch.empty(Mrows, Vp, dtype=torch.float32, device=DEV)
    K.gemm(xd, wd, logits, M=Mrows, N=V, K=256, lda=256, ldb=256, ldc=Vp, bias=bd)
    idx2 = torch.empty(Mrows, dtype=torch.int32, device=DEV)
    top2 = torch.empty(Mrows, dtype=torch.float32, device=DEV)
    lse2 = torch.empty(Mrows, dtype=torch.float32, device=DEV)
    K.argmax_lse(logits, Vp, Mrows, V, idx2, top2, lse2)
    torch.cuda.synchronize()
    n_diff = int((idx2 != idx).sum())
    assert n_diff <= max(2, Mrows // 2000), n_diff   # (two fp32 summation orders: near-ties only)
    np.testing.assert_allclose(lse.cpu().numpy(), lse2.cpu().numpy(), rtol=0, atol=2e-4)


def test_head_greedy_live_row_bound_and_row_stride():
    """Only rows below the live count are written (a packed batch); x may be a strided view (row stride 512)."""
    g = torch.Generator().manual_seed(3)
    Mrows, V = 300, 512
    wide = torch.randn(Mrows, 512, generator=g).bfloat16().to(DEV)
    x = wide[:, :256]
    w = (torch.randn(V, 256, generator=g) * 0.06).bfloat16().to(DEV)
    idx = torch.full((Mrows,), -5, dtype=torch.int32, device=DEV)
    top = torch.full((Mrows,), 7.0, device=DEV)
    K.ctc_head_greedy(x, w, None, idx, top, None)
    ref = (x.float() @ w.float().t())
    assert int((idx.long() != ref.argmax(1)).sum()) <= 1
    # live bound through a packed geometry
    from s2t_amd import rows as Rows
    lens = torch.tensor([100, 60, 40], dtype=torch.int32, device=DEV)
    Rows.attach(lens, 3, 100, 0, tag=("test_ctc_head",))
    live = Rows.K.rows_geom(lens).live_rows()
    assert live == 200
    idx.fill_(-5)
    K.ctc_head_greedy(x, w, None, idx, top, None, bound=lens)
    torch.cuda.synchronize()
    assert int((idx[:live] >= 0).sum()) == live and int((idx[live:] == -5).sum()) == Mrows - live


def test_greedy_decode_with_the_fused_head_equals_the_two_kernel_decode():
    """CTCDecoder.generate on configuration 5a's model (12-layer Conformer + CTC head, bf16, fp32 CTC logits) at 48 x 1000: the fused
    head (no logits) against the same decoder with S2T_CTC_HEAD_FUSED off: the same hypotheses up to frames whose top-2 logits tie
    to fp32 rounding; scores to 1e-3."""
    import bench

    V = 10000
    torch.manual_seed(1)
    a = M.recipe_args(conformer=True, vocab_size=V, ctc_weight=1.0)
    model = M.S2TCTCModel.build_model(a, M.FakeTask(V)).prepare(torch.bfloat16, DEV)
    model.encoder.ctc_out_dtype = torch.float32
    model.eval()
    sample = bench.synthetic_batch(48, 1000, V, 5, torch.device(DEV))[0]
    dec = M.CTCDecoder([model], None, None)
    assert dec.fused_head
    calls = []
    orig = K.ctc_head_greedy

    def spy(*a_, **k_):
        calls.append(1)
        return orig(*a_, **k_)

    K.ctc_head_greedy = spy
    try:
        with torch.no_grad():
            fused = dec.generate([model], sample)
            assert len(calls) == 1, "the fused head did not run"
            dec.fused_head = False
            plain = dec.generate([model], sample)
            assert len(calls) == 1
    finally:
        K.ctc_head_greedy = orig
    assert not getattr(model.encoder, "ctc_greedy_only", False)   # the flag is the decode's, not the model's
    n_tok = sum(len(h[0]["tokens"]) for h in plain)
    assert n_tok > 0
    differ = [b for b in range(len(plain)) if fused[b][0]["tokens"].tolist() != plain[b][0]["tokens"].tolist()]
    assert len(differ) <= 2, differ
    for b in range(len(plain)):
        if b not in differ:
            assert abs(float(fused[b][0]["score"]) - float(plain[b][0]["score"])) <= 1e-3 * max(1.0, abs(float(plain[b][0]["score"])))
