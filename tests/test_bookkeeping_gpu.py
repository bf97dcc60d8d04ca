"""Per-batch bookkeeping in one launch per item (csrc/bookkeeping.hip) against the ATen compositions it replaces — the
reference's own arithmetic: subsampling.py:150-154 + data_utils.py:518-522 (lengths, padding mask), utils.py:240-250
(make_positions), criterions/ctc.py:516-540 (CTC targets) — bit for bit (integer work), and a captured step whose batches are
refreshed through these entries against eager steps that compose everything from ATen ops."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import functional as Fn  # noqa: E402
from s2t_amd import kernels as K  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402
from s2t_amd import trainer as TR  # noqa: E402
from s2t_amd.modules import Conv1dSubsampling  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("B,T", [(1, 7), (64, 1000), (33, 2001), (5, 1)])
def test_subsampled_lengths_and_mask(B, T):
    g = torch.Generator().manual_seed(B * 1000 + T)
    sl = torch.randint(1, T + 1, (B,), generator=g).to(DEV)
    sl[0] = T
    Tp = ((T - 1) // 2 + 1 - 1) // 2 + 1
    ref = Conv1dSubsampling.get_out_seq_lens_tensor(sl)
    ref_mask = torch.arange(Tp, device=DEV)[None, :] >= ref[:, None]
    l64 = torch.full((B,), -7, dtype=torch.int64, device=DEV)
    l32 = torch.full((B,), -7, dtype=torch.int32, device=DEV)
    mask = torch.zeros(B, Tp, dtype=torch.bool, device=DEV)
    K.subsampled_lengths(sl, Tp, l64, l32, mask)
    assert torch.equal(l64, ref) and torch.equal(l32, ref.to(torch.int32)) and torch.equal(mask, ref_mask)


@pytest.mark.parametrize("B,U,pad", [(64, 61, 1), (3, 200, 1), (7, 1, 0), (16, 64, 1), (16, 65, 1)])
def test_token_positions(B, U, pad):
    g = torch.Generator().manual_seed(B * 100 + U)
    tok = torch.randint(2, 50, (B, U), generator=g)
    n = torch.randint(0, U + 1, (B,), generator=g)
    n[0] = U
    if B > 1:
        n[1] = 0
    for b in range(B):
        tok[b, int(n[b]):] = pad
    if U > 5 and B > 2:
        tok[2, 1] = pad  # a pad in the middle (the reference's arithmetic covers it)
    tok = tok.to(DEV)
    nonpad = tok.ne(pad)
    ref_pos = (torch.cumsum(nonpad, dim=1) * nonpad + pad).to(torch.int32)
    pos = torch.full((B, U), -9, dtype=torch.int32, device=DEV)
    cnt = torch.full((B,), -9, dtype=torch.int32, device=DEV)
    K.token_positions(tok, pad, pos, cnt)
    assert torch.equal(pos, ref_pos) and torch.equal(cnt, nonpad.sum(1).to(torch.int32))


@pytest.mark.parametrize("B,U", [(64, 61), (3, 200), (5, 1), (9, 64), (9, 129)])
def test_ctc_targets(B, U):
    g = torch.Generator().manual_seed(B * 100 + U + 1)
    pad, eos = 1, 2
    t = torch.randint(3, 40, (B, U), generator=g)
    n = torch.randint(0, U, (B,), generator=g)
    for b in range(B):
        t[b, int(n[b])] = eos
        t[b, int(n[b]) + 1:] = pad
    if U > 6 and B > 2:
        t[2, 0] = eos  # an eos in front of labels
        t[2, 3] = pad
    t = t.to(DEV)
    ref_mat, ref_len = C.ctc_targets(t, pad, eos)
    tm = torch.full((B, U), -5, dtype=torch.int64, device=DEV)
    tl = torch.full((B,), -5, dtype=torch.int32, device=DEV)
    K.ctc_targets(t, pad, eos, tm, tl)
    assert torch.equal(tm, ref_mat) and torch.equal(tl, ref_len)


def test_gather_rows_through_a_row_map():
    B, U = 37, 61
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(0, U + 1, (B,), generator=g).to(torch.int32).to(DEV)
    src = torch.randint(0, 1000, (B * U,), generator=g).to(DEV)
    cu = torch.empty(B + 1, dtype=torch.int32, device=DEV)
    buf = torch.empty(4 + B * U, dtype=torch.int32, device=DEV)
    K.rows_geometry(lens, B, U, 0, cu, buf)
    m = buf[4:]
    ok = m >= 0
    idx = torch.where(ok, (m >> 16).long() * U + (m & 0xffff).long(), torch.zeros_like(m, dtype=torch.long))
    ref = torch.where(ok, src[idx], torch.full_like(idx, 1))
    out = torch.full((B * U,), -3, dtype=torch.int64, device=DEV)
    K.gather_rows_i64(src, m, U, 1, out)
    assert torch.equal(out, ref)


def test_refresh_through_the_one_launch_forms_equals_the_aten_composition():
    """Trainer.load_batch refreshes the bookkeeping of a captured step through the entries above (functional._recompute_in_place
    takes a memo's ``into`` form); with the forms removed it recomputes from ATen ops and copies.  Same batches, same losses,
    and the refreshed tensors themselves are equal."""
    V = 200

    def sample(seed, B=24, T=1000, U=31):
        g = torch.Generator().manual_seed(seed)
        sl = torch.randint(T // 2, T + 1, (B,), generator=g)
        sl[0] = T
        src = torch.randn(B, T, 80, generator=g)
        tl = torch.randint(5, U, (B,), generator=g)
        tl[0] = U - 1
        tgt = torch.full((B, U), 1, dtype=torch.int64)
        prev = torch.full((B, U), 1, dtype=torch.int64)
        for b in range(B):
            n = int(tl[b])
            lab = torch.randint(4, V, (n,), generator=g)
            tgt[b, :n] = lab
            tgt[b, n] = 2
            prev[b, 0] = 2
            prev[b, 1:n + 1] = lab
        return {"net_input": {"src_tokens": src.to(DEV), "src_lengths": sl.to(DEV), "prev_output_tokens": prev.to(DEV)},
                "target": tgt.to(DEV), "ntokens": int(tl.sum()) + B}

    batches = [sample(11 + i) for i in range(3)]

    def run(strip):
        torch.manual_seed(0)
        args = M.recipe_args(conformer=True, vocab_size=V, encoder_layers=2, decoder_layers=1, dropout=0.0, attention_dropout=0.0,
                             activation_dropout=0.0)
        model = M.S2TTransformerModel.build_model(args, M.FakeTask(V)).prepare(torch.bfloat16, torch.device(DEV))
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        tr = TR.Trainer(model, crit, lr=1e-5, warmup_updates=1, clip_norm=10.0)
        first = {"net_input": {k: v.clone() for k, v in batches[0]["net_input"].items()}, "target": batches[0]["target"].clone(),
                 "ntokens": batches[0]["ntokens"]}
        tr.capture(first)
        if strip:
            n = 0
            for _, _, _, e in Fn._all_memo_entries():
                if getattr(e[2], "into", None) is not None:
                    del e[2].into
                    n += 1
            assert n >= 3  # lengths, tokens, targets (and the packed targets where the decoder packs)
        losses = [float(tr.replay(bt)[0]) for bt in batches[1:] + batches[:1]]
        memo = {}
        for _, key, _, e in Fn._all_memo_entries():
            if isinstance(key, tuple) and key and key[0] in ("enc_lens", "dec_tokens", "targets", "targets_packed"):
                memo[key[0]] = [o.clone() for o in e[3] if torch.is_tensor(o)]
        tr.release()
        return losses, memo

    la, ma = run(False)
    lb, mb = run(True)
    # The bookkeeping tensors are compared bit for bit below.  The LOSSES agree to rounding, not always to the bit: the per-parameter
    # gradient sums (LayerNorm gains, biases) are fp32 atomics, Adam carries a last-bit difference into the masters, and once in a
    # few runs one bf16 shadow weight rounds the other way — the third loss then moves by ~1e-5 (seen on MI355X in round 6 with the
    # round-5 library as well: 1 of 6 runs).
    assert np.all(np.isfinite(la))
    np.testing.assert_allclose(la, lb, rtol=2e-4)
    assert set(ma) == set(mb) and len(ma) >= 3
    for k in ma:
        assert len(ma[k]) == len(mb[k]) and all(torch.equal(a, b) for a, b in zip(ma[k], mb[k])), k
