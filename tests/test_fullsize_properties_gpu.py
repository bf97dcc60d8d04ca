"""Size-independent properties at the BASELINE.json configuration-2' size (12-layer Conformer, 64 x 1000 x 80, bf16,
V = 10 000), where the CPU oracle is too slow to serve as the checker:

  * utterances are independent in eval mode: permuting the batch permutes the outputs BIT-EXACTLY (every GEMM row,
    attention (utterance, head) and depthwise-conv column is computed by the same arithmetic wherever it sits);
  * padding invariance: appending zero frames beyond every utterance's length changes no CTC-greedy token id;
  * the summed loss is additive over utterances (eval-mode BatchNorm, dropout 0);
  * the grouped weight-gradient launch, the per-weight split-K GEMMs and (through the trainer test) graph replay give
    the same gradients.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402
from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import functional as Fn  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402

DEV = torch.device("cuda", 0)
V = 10000
B, T = 64, 1000


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(1)
    m = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V), M.FakeTask(V)).prepare(torch.bfloat16, DEV)
    m.encoder.ctc_out_dtype = torch.float32
    return m


@pytest.fixture(scope="module")
def sample():
    return bench.synthetic_batch(B, T, V, 7, DEV)[0]


class _EncOnly(torch.nn.Module):
    def __init__(self, e):
        super().__init__()
        self.e = e

    def forward(self, src_tokens, src_lengths):
        return self.e(src_tokens, src_lengths)


def _greedy(model, src, lens):
    dec = M.CTCDecoder([model.encoder], None, None)
    dec.model = _EncOnly(model.encoder)
    with torch.no_grad():
        hyps = dec.generate(None, {"net_input": {"src_tokens": src, "src_lengths": lens}})
    return [h[0]["tokens"].tolist() for h in hyps]


def test_batch_permutation_is_bit_exact(model, sample):
    model.eval()
    ni = sample["net_input"]
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        a = model.encoder(ni["src_tokens"], ni["src_lengths"])
        b = model.encoder(ni["src_tokens"][perm].contiguous(), ni["src_lengths"][perm].contiguous())
    assert torch.equal(a["encoder_out"][0][:, perm], b["encoder_out"][0])
    assert torch.equal(a["ctc_logit"][0][:, perm], b["ctc_logit"][0])
    ids = _greedy(model, ni["src_tokens"], ni["src_lengths"])
    ids_p = _greedy(model, ni["src_tokens"][perm].contiguous(), ni["src_lengths"][perm].contiguous())
    assert [ids[i] for i in perm.tolist()] == ids_p
    assert sum(len(x) for x in ids) > 0


def test_extra_padding_changes_no_token(model, sample):
    """Holds for utterances that end at least a few frames before the batch's T: like the reference's Conv1dSubsampling
    (no mask between its two convolutions, subsampling.py:145-159), the second convolution of a FULL-length utterance
    sees the zero padding of the buffer edge where a longer buffer holds GLU(bias) of the padded frames.
    The two buffers have different row counts (16 000 and 16 640): the fused feed-forward kernel would run the first with two
    workgroups per row block and the second with one, which sum the hidden units' products in different orders; bit-equal ids
    across row counts are a property of ONE summation order, so it is pinned here (S2T_FFN_PC_SPLIT, csrc/rowblock.hip)."""
    from s2t_amd import kernels as K

    _, old, _ = K.ffn_configure()
    K.ffn_configure(split=1)
    try:
        model.eval()
        ni = sample["net_input"]
        src = ni["src_tokens"].clone()
        lens = ni["src_lengths"].clamp(max=T - 16)
        for b in range(B):
            src[b, int(lens[b]):] = 0
        padded = torch.zeros(B, T + 40, 80, device=DEV)
        padded[:, :T] = src
        assert _greedy(model, src, lens) == _greedy(model, padded, lens)
    finally:
        K.ffn_configure(split=old)


def test_extra_padding_in_the_default_configuration(model, sample):
    """The same batch under the SHIPPED switches (the fused feed-forward kernels deal a row block's hidden units to as many
    workgroups as the row count leaves CUs for; packed rows put an utterance at another row offset when the padded length
    moves its halo): bit equality is a property of one summation order (the test above), what holds here is that the CTC
    logits of every frame agree to bf16 rounding noise and the greedy ids with them."""
    model.eval()
    ni = sample["net_input"]
    src = ni["src_tokens"].clone()
    lens = ni["src_lengths"].clamp(max=T - 16)
    for b in range(B):
        src[b, int(lens[b]):] = 0
    padded = torch.zeros(B, T + 40, 80, device=DEV)
    padded[:, :T] = src
    with torch.no_grad():
        a = model.encoder(src, lens)["ctc_logit"][0].float()      # T' x B x V
        b = model.encoder(padded, lens)["ctc_logit"][0].float()
    sub = model.encoder.subsample.get_out_seq_lens_tensor(lens.cpu())
    valid = (torch.arange(a.shape[0])[:, None] < sub[None, :]).to(DEV)
    xa, xb = a[valid], b[:a.shape[0]][valid]
    assert float((xa - xb).norm() / xa.norm()) <= 1.5e-2  # (measured 0.0066: last-bit flips through twelve bf16 layers)
    # (per frame, not per collapsed token: one flipped frame shifts every later token of its utterance; the random weights of
    # this model leave a few per cent of the frames with near-tied logits)
    agree = float((xa.argmax(-1) == xb.argmax(-1)).float().mean())
    assert agree >= 0.97, agree


def test_loss_is_additive_over_utterances(model, sample):
    model.eval()  # BatchNorm with running statistics: no coupling between utterances
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    ni = sample["net_input"]
    with torch.no_grad():
        full = float(crit(model, sample)[0])
        parts = 0.0
        for i in range(0, B, 16):
            sub = {"net_input": {k: v[i:i + 16].contiguous() for k, v in ni.items()}, "target": sample["target"][i:i + 16].contiguous(),
                   "ntokens": 1}
            parts += float(crit(model, sub)[0])
    assert abs(full - parts) <= 2e-3 * abs(full), (full, parts)


def test_grouped_and_per_weight_gradients_agree(sample):
    """On the 12-layer Transformer (config 2): its data path has no order-dependent reduction, so two identical passes
    give identical activation gradients and the two weight-gradient paths differ only in their fp32 summation trees."""
    torch.manual_seed(2)
    model = M.S2TTransformerModel.build_model(M.recipe_args(conformer=False, vocab_size=V), M.FakeTask(V)).prepare(torch.bfloat16, DEV)
    model.train()
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    grads = {}
    old = Fn._WGQ["mode"]
    try:
        for tag, mode in (("grouped", "1"), ("per_weight", "0"), ("per_weight_again", "0")):
            Fn._WGQ["mode"] = mode
            model.flat.zero_grad()
            loss, _, _ = crit(model, sample)
            loss.backward()
            torch.cuda.synchronize()
            grads[tag] = model.flat.grad.clone()
    finally:
        Fn._WGQ["mode"] = old
    ref = grads["per_weight"]
    assert torch.isfinite(ref).all() and torch.isfinite(grads["grouped"]).all()
    noise = float((grads["per_weight_again"] - ref).norm() / ref.norm())  # LayerNorm / bias partial sums use atomics
    assert noise < 1e-4, noise
    assert float((grads["grouped"] - ref).norm() / ref.norm()) < 1e-3


def test_grouped_weight_gradients_split_by_kernel(sample):
    """A backward pass whose queue holds problems the 256 x 256 LDS-DMA kernel cannot take (an unaligned leading dimension, an
    operand beyond 2 GiB: functional.wgrad256_eligible) runs them on the 128 x 128 grouped kernel in a second launch and the
    rest on the large tiles — it no longer moves the whole pass to the slow kernel or aborts.  Rehearsed by
    declaring every problem with a 2048-wide operand ineligible: eager and inside a captured step (two staging blocks per flush),
    gradients equal to the all-eligible pass up to the two kernels' summation trees; tied weights (the CTC head and the decoder's
    embedding / output projection) stay in ONE launch."""
    from s2t_amd import rows as Rows
    from s2t_amd.trainer import Trainer

    old_rows, Rows.ENABLED = Rows.ENABLED, False  # (padded rows: a packed problem has no second kernel to go to)
    old_mode, old_fn = Fn._WGQ["mode"], Fn.wgrad256_eligible
    launches = []
    orig_group = Fn._flush_wgrad_group

    def spy(q, big):
        launches.append((len(q), big))
        return orig_group(q, big)

    def picky(M_, ldy, ldx, py=0, px=0):
        return old_fn(M_, ldy, ldx, py, px) and ldy != 2048 and ldx != 2048

    try:
        Fn._WGQ["mode"] = "1"
        Fn._flush_wgrad_group = spy
        grads = {}
        for tag, fn in (("all256", old_fn), ("split", picky)):
            torch.manual_seed(2)
            model = M.S2TTransformerModel.build_model(M.recipe_args(conformer=False, vocab_size=V, encoder_layers=3, decoder_layers=2),
                                                      M.FakeTask(V)).prepare(torch.bfloat16, DEV)
            model.train()
            crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
            Fn.wgrad256_eligible = fn
            launches.clear()
            model.flat.zero_grad()
            loss, _, _ = crit(model, sample)
            loss.backward()
            torch.cuda.synchronize()
            grads[tag] = (model.flat.grad.clone(), list(launches))
            if tag == "split":  # the same split inside a captured step: two launches per flush, each with a staging block of its own
                tr = Trainer(model, crit, lr=1e-6, warmup_updates=1)
                tr.capture(sample)
                losses = [float(tr.replay()[0]) for _ in range(3)]
                assert all(torch.isfinite(torch.tensor(losses))), losses
                tr.release()
        assert [b for _, b in grads["all256"][1]] == [True]
        assert sorted(b for _, b in grads["split"][1]) == [False, True], grads["split"][1]
        ref, got = grads["all256"][0], grads["split"][0]
        assert torch.isfinite(got).all()
        assert float((got - ref).norm() / ref.norm()) < 1e-3
    finally:
        Fn._WGQ["mode"], Fn.wgrad256_eligible, Fn._flush_wgrad_group, Rows.ENABLED = old_mode, old_fn, orig_group, old_rows


def test_conformer_training_pass_repeats(sample):
    """The Conformer's training-mode data path holds no order-dependent reduction either: BatchNorm batch statistics and
    the BatchNorm backward sums are per-workgroup partial rows added in a fixed order (conv.hip), so two identical passes
    give the same loss and the same activation gradients; what is left is the fp32 atomics in the
    per-parameter sums (LayerNorm dgamma/dbeta folds, bias column sums)."""
    torch.manual_seed(2)
    model = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V), M.FakeTask(V)).prepare(torch.bfloat16, DEV)
    model.train()
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    out = []
    for _ in range(2):
        model.flat.zero_grad()
        loss, _, _ = crit(model, sample)
        loss.backward()
        torch.cuda.synchronize()
        out.append((float(loss.detach()), model.flat.grad.clone()))
    noise = float((out[1][1] - out[0][1]).norm() / out[0][1].norm())
    print("loss", out[0][0], out[1][0], "grad noise", noise)
    assert abs(out[0][0] - out[1][0]) <= 2e-6 * abs(out[0][0])  # the loss scalar itself is an atomic sum over rows
    assert noise < 1e-4, noise


@pytest.mark.parametrize("packed", [False, True])
def test_one_pass_relpos_backward_equals_the_three_kernel_route_in_the_model(sample, packed):
    """The bench's Conformer at its literal batch: a training pass whose relative-position attention backward runs in ONE launch per
    layer (s2t_relpos_attn_bwd, T' = 250 <= 256) against the same pass on the three-kernel route (s2t_attn_fused_bwd writing the
    skewed score gradient + s2t_relpos_glue; S2T_RELPOS_ONE_PASS=0) — same loss (the forward is the same code), every parameter
    gradient equal up to the two routes' roundings (the one-pass kernel rounds dq once, where the glue added into a bf16 dq)."""
    from s2t_amd import rows as Rows

    old_rows, old_one = Rows.ENABLED, Fn._RELPOS_ONE_PASS
    res = {}
    try:
        Rows.ENABLED = packed
        for one in (False, True):
            Fn._RELPOS_ONE_PASS = one
            torch.manual_seed(2)
            model = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V, encoder_layers=4, decoder_layers=2),
                                                      M.FakeTask(V)).prepare(torch.bfloat16, DEV)
            model.train()
            crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
            model.flat.zero_grad()
            loss, _, _ = crit(model, sample)
            loss.backward()
            torch.cuda.synchronize()
            res[one] = (float(loss.detach()), {k: p.grad.detach().float().clone() for k, p in model.named_parameters()})
    finally:
        Rows.ENABLED, Fn._RELPOS_ONE_PASS = old_rows, old_one
    assert abs(res[False][0] - res[True][0]) <= 2e-6 * abs(res[False][0]), (res[False][0], res[True][0])
    errs = {}
    for k, ga in res[False][1].items():
        den = float(ga.norm())
        if den < 1e-6 or k.endswith(("linear_k.bias", "k_proj.bias")):  # (mathematically zero: softmax is shift invariant)
            continue
        errs[k] = float((ga - res[True][1][k]).norm()) / den
    worst = max(errs, key=errs.get)
    print("one-pass vs three-kernel route (packed=%s): median %.2e, worst %.2e (%s)" % (packed, float(np.median(list(errs.values()))), errs[worst], worst))
    assert errs[worst] <= 3e-2, (worst, errs[worst])
    assert float(np.median(list(errs.values()))) <= 5e-3
