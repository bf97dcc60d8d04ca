"""Packed rows (s2t_amd/rows.py; include/s2t_hip.h "Packed rows") against the padded layout of the SAME bf16 HIP model.

The reference computes on the padded frames of a batch (data/audio/speech_to_text_dataset.py:411-485,
models/speech_to_text/s2t_transformer.py:1765-1946, modules/convolution.py:76-120); the packed layout skips them.  What must
not move: every frame's encoder output, CTC logit and decoder logit, the greedy token ids, the joint loss, every parameter
gradient and the BatchNorm running statistics (halo rows: the padded frames whose depthwise-convolution output is not zero).
The padded path is the one the golden fixtures and the oracle pin (tests/test_model_parity_gpu.py, test_configs_fullsize_gpu.py),
so equality with it carries their parity over.  Row-local arithmetic is the same instructions on the same values in both
layouts: eval outputs are compared bit for bit; training differs only in the order of the cross-row sums (BatchNorm statistics,
weight gradients), bounded tightly.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import functional as Fn  # noqa: E402
from s2t_amd import kernels as K  # noqa: E402
from s2t_amd import rows as Rows  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402
from s2t_amd import trainer as TR  # noqa: E402

DEV = "cuda"
V = 1000


def _sample(B, T, seed, full_first=True, lo=0.55):
    g = torch.Generator().manual_seed(seed)
    lens = [T] if full_first else []
    lens += [int(torch.randint(int(lo * T), T + 1, (1,), generator=g)) for _ in range(B - len(lens))]
    lens = sorted(lens, reverse=True)
    src = torch.randn(B, T, 80, generator=g)
    for b, l in enumerate(lens):
        src[b, l:] = 0
    ul = [int(torch.randint(30, 91, (1,), generator=g)) for _ in range(B)]  # (24 x 91 target rows: enough for the decoder's
    U = 91                                                                   #  row-block kernels, so its rows run packed too)
    target = torch.full((B, U), 1, dtype=torch.long)
    prev = torch.full((B, U), 1, dtype=torch.long)
    for b, u in enumerate(ul):
        toks = torch.randint(4, V, (u,), generator=g)
        target[b, :u] = toks
        target[b, u] = 2
        prev[b, 0] = 2
        prev[b, 1:u + 1] = toks
    return {"net_input": {"src_tokens": src.to(DEV), "src_lengths": torch.tensor(lens).to(DEV),
                          "prev_output_tokens": prev.to(DEV)},
            "target": target.to(DEV), "ntokens": int(sum(ul) + B)}, lens


def _model(conformer, enc_layers=3, dec_layers=2, seed=0, dropout=0.0):
    torch.manual_seed(seed)
    a = M.recipe_args(conformer=conformer, vocab_size=V, encoder_layers=enc_layers, decoder_layers=dec_layers)
    a.dropout = a.attention_dropout = a.activation_dropout = dropout
    model = M.S2TTransformerModel.build_model(a, M.FakeTask(V))
    g = torch.Generator().manual_seed(seed + 7)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for n_, b in model.named_buffers():
            if n_.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            if n_.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
    model.prepare(torch.bfloat16, DEV)
    return model


class _layout:
    def __init__(self, packed):
        self.packed = packed

    def __enter__(self):
        self.old, Rows.ENABLED = Rows.ENABLED, self.packed

    def __exit__(self, *exc):
        Rows.ENABLED = self.old


def test_row_map_and_round_trip():
    B, T, halo = 5, 37, 7
    lens = torch.tensor([37, 35, 30, 12, 1], dtype=torch.int32, device=DEV)
    Rows.attach(lens, B, T, halo)
    g = lens._pk
    cap = [min(l + halo, T) for l in lens.tolist()]
    assert g.cu.tolist() == [0] + list(np.cumsum(cap))
    assert g.live_rows() == sum(cap)
    m = g.row_map.tolist()
    r = 0
    for b, (l, c) in enumerate(zip(lens.tolist(), cap)):
        for t in range(c):
            assert m[r] == ((b << 16) | t if t < l else -1), (b, t)
            r += 1
    assert all(v == -1 for v in m[r:])
    x = torch.randn(B * T, 64, device=DEV).bfloat16()
    valid = (torch.arange(T, device=DEV)[None, :] < lens[:, None]).reshape(-1)
    xp = Rows.pack(x, lens)
    back = Rows.unpack(xp, lens)
    assert torch.equal(back[valid], x[valid]) and float(back[~valid].float().abs().max()) == 0.0
    # halo rows of the packed matrix are zero
    halo_rows = torch.tensor([i for i in range(sum(cap)) if m[i] < 0], device=DEV, dtype=torch.long)
    assert float(xp[halo_rows].float().abs().max()) == 0.0


@pytest.mark.parametrize("B", [1, 1023, 1024])
def test_row_geometry_at_the_largest_batch_of_the_one_launch_form(B):
    """s2t_rows_geometry serves up to 1024 utterances with one 1024-thread workgroup scan: cu[B] (the live row count, which
    bounds the LAST utterance for every per-utterance kernel) has no thread of its own when B = 1024."""
    T, halo = 9, 2
    g_ = torch.Generator().manual_seed(B)
    lens = torch.randint(1, T + 1, (B,), generator=g_).to(torch.int32)
    cu = torch.full((B + 1,), -7, dtype=torch.int32, device=DEV)  # poisoned: an unwritten entry shows
    buf = torch.full((Rows.PackedRows.HEADER + B * T,), -7, dtype=torch.int32, device=DEV)
    K.rows_geometry(lens.to(DEV), B, T, halo, cu, buf)
    cap = [min(l + halo, T) for l in lens.tolist()]
    assert cu.tolist() == [0] + list(np.cumsum(cap))
    assert int(buf[Rows.PackedRows.HEADER - 1]) == sum(cap)
    m = buf[Rows.PackedRows.HEADER:].tolist()
    exp = []
    for b, (l, c) in enumerate(zip(lens.tolist(), cap)):
        exp += [((b << 16) | t) if t < l else -1 for t in range(c)]
    assert m[:len(exp)] == exp and all(v == -1 for v in m[len(exp):])


@pytest.mark.parametrize("conformer", [True, False])
@pytest.mark.parametrize("split", [1, 0])
def test_eval_outputs_equal_the_padded_layout(conformer, split):
    """Encoder output, CTC logits and decoder logits on every frame / token, and the greedy ids.  With one workgroup per row
    block of the fused feed-forward kernels (split = 1) bit for bit: a frame's arithmetic does not depend on the row it sits
    in.  In the default configuration (split = 0: the hidden units of a row block dealt to several workgroups for the few
    thousand rows of this test) the fp32 partial rows of a frame are added in an order that follows its place in the block —
    last-bit differences of single frames, bounded here."""
    model = _model(conformer)
    model.eval()
    sample, lens = _sample(24, 1000, 3)
    ni = sample["net_input"]
    outs = {}
    _, old_split, _ = K.ffn_configure()
    K.ffn_configure(split=split)
    try:
        outs = _eval_both(model, sample, ni)
    finally:
        K.ffn_configure(split=old_split)
    Tp = outs[False][0].shape[0]
    sub = model.encoder.subsample.get_out_seq_lens_tensor(torch.tensor(lens))
    valid = (torch.arange(Tp)[:, None] < sub[None, :]).to(DEV)  # T x B
    if split == 1:
        for i in (0, 1):
            a, b = outs[False][i], outs[True][i]
            assert a.shape == b.shape
            assert torch.equal(a[valid], b[valid]), "frames differ between the layouts (output %d)" % i
        tokv = ni["prev_output_tokens"] != 1  # (the target rows run packed too: padded positions hold no token)
        assert torch.equal(outs[False][2][tokv], outs[True][2][tokv]), "decoder logits differ"
        assert outs[False][3] == outs[True][3], "greedy ids differ"
    else:
        for i in (0, 1):
            a, b = outs[False][i][valid], outs[True][i][valid]
            assert float((a - b).norm() / a.norm()) <= 5e-3, i
        tokv = ni["prev_output_tokens"] != 1
        a, b = outs[False][2][tokv], outs[True][2][tokv]
        assert float((a - b).norm() / a.norm()) <= 5e-3
        # (greedy ids: compared in the split = 1 run only — the random weights of this model leave many frames with near-tied
        # logits, which a last-bit difference tips either way in any layout)
    assert any(len(h) > 0 for h in outs[True][3])


def _sample_with_lens(lens, T, seed):
    smp, _ = _sample(len(lens), T, seed)
    src = smp["net_input"]["src_tokens"]
    for b, l in enumerate(lens):
        src[b, l:] = 0
    smp["net_input"]["src_lengths"] = torch.tensor(lens).to(DEV)
    return smp


@pytest.mark.parametrize("case", ["all_full", "tiny_utterances"])
def test_edge_fills_equal_the_padded_layout(case):
    """No padding at all (every utterance full length: cu is the uniform layout, no halo row) and utterances of a handful of
    frames (shorter than the depthwise kernel's reach, than an attention tile, than their CTC target allows): eval outputs bit
    for bit with one workgroup per FFN row block, one training pass within the summation-order spread."""
    T = 1000
    lens = [T] * 24 if case == "all_full" else sorted([T, 997, 640, 333, 130, 41, 29, 17, 9, 5, 3, 1] * 2, reverse=True)
    sample = _sample_with_lens(lens, T, 31)
    ni = sample["net_input"]
    model = _model(True)
    model.eval()
    _, old_split, _ = K.ffn_configure()
    K.ffn_configure(split=1)
    try:
        outs = _eval_both(model, sample, ni)
    finally:
        K.ffn_configure(split=old_split)
    sub = model.encoder.subsample.get_out_seq_lens_tensor(torch.tensor(lens))
    Tp = outs[False][0].shape[0]
    valid = (torch.arange(Tp)[:, None] < sub[None, :]).to(DEV)
    for i in (0, 1):
        assert torch.equal(outs[False][i][valid], outs[True][i][valid]), i
    tokv = ni["prev_output_tokens"] != 1
    assert torch.equal(outs[False][2][tokv], outs[True][2][tokv]) and outs[False][3] == outs[True][3]
    res = {}
    for packed in (False, True):
        m = _model(True)
        m.train()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        with _layout(packed):
            m.flat.zero_grad()
            loss, _, log = crit(m, sample)
            loss.backward()
            torch.cuda.synchronize()
        res[packed] = (float(loss.detach()), {k: p.grad.detach().float().clone() for k, p in m.named_parameters()})
    assert np.isfinite(res[True][0]) and abs(res[False][0] - res[True][0]) <= 5e-4 * abs(res[False][0]), (res[False][0], res[True][0])
    errs = []
    for k, ga in res[False][1].items():
        den = float(ga.norm())
        if k.endswith(("k_proj.bias", "linear_k.bias")) or den < 1e-5:
            continue
        gb = res[True][1][k]
        assert bool(torch.isfinite(gb).all()), k
        errs.append(float((ga - gb).norm()) / den)
    assert max(errs) <= 0.1 and float(np.median(errs)) <= 0.02, (max(errs), float(np.median(errs)))


def test_long_utterances_under_relative_positions():
    """More than 256 frames per utterance with relative positions (the PDS stages run 1004 / 502): the backward behind the skewed
    score gradient — s2t_relpos_glue — walks the position rows in chunks of 512 since round 5, so training runs packed at any
    length.  Inference: the packed ids are the padded layout's; training: loss and gradients agree with the padded layout within
    the spread of two summation orders (the padded layout takes the same kernel, unpacked)."""
    sample, lens = _sample(12, 1500, 41)
    ni = sample["net_input"]
    ids = {}
    res = {}
    _, old_split, _ = K.ffn_configure()
    for packed in (False, True):
        model = _model(True, enc_layers=2, dec_layers=1)
        model.eval()
        K.ffn_configure(split=1)  # (one summation order of the fused feed-forward kernels: see test_eval_outputs_…)
        try:
            with torch.no_grad(), _layout(packed):
                enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
                assert (enc.get("packed") is not None) == packed
                ids[packed] = [h[0]["tokens"].tolist() for h in M.CTCDecoder([model.encoder]).generate([model.encoder], sample)]
        finally:
            K.ffn_configure(split=old_split)
        model.train()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        with _layout(packed):
            enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
            assert (enc.get("packed") is not None) == packed  # T' = 375 > 256: packed in training too
            model.flat.zero_grad()
            loss, _, _ = crit(model, sample)
            loss.backward()
            torch.cuda.synchronize()
        res[packed] = (float(loss.detach()), {k: p.grad.detach().float().clone() for k, p in model.named_parameters()})
    assert ids[False] == ids[True]
    assert np.isfinite(res[True][0]) and abs(res[False][0] - res[True][0]) <= 5e-4 * abs(res[False][0]), (res[False][0], res[True][0])
    errs = []
    for k, ga in res[False][1].items():
        den = float(ga.norm())
        if k.endswith(("k_proj.bias", "linear_k.bias")) or den < 1e-5:
            continue
        e = float((ga - res[True][1][k]).norm()) / den
        errs.append(e)
        assert e <= 0.08, (k, e)
    assert float(np.median(errs)) <= 0.02


def test_beam_search_over_a_packed_encoder():
    """fairseq's SequenceGenerator (sequence_generator.py:191-786) reads the encoder dictionary's T x B x C entries and re-orders
    them per beam: with packed rows those entries are materialised on first access (rows.LazyList).  Plain search and the joint
    CTC / attention search (the prefix scorer reads ``ctc_logit``) give the padded layout's hypotheses."""
    from s2t_amd.sequence_generator import SequenceGenerator

    model = _model(True, enc_layers=2, dec_layers=1)
    model.eval()
    sample, _ = _sample(20, 1000, 51)
    task = M.FakeTask(V)
    _, old_split, _ = K.ffn_configure()
    K.ffn_configure(split=1)
    try:
        for ctc_w in (0.0, 0.3):
            out = {}
            for packed in (False, True):
                with _layout(packed):
                    gen = SequenceGenerator([model], task.target_dictionary, beam_size=3, max_len_a=0, max_len_b=12, ctc_weight=ctc_w)
                    hyps = gen.generate([model], {"net_input": {k: sample["net_input"][k] for k in ("src_tokens", "src_lengths")}})
                    out[packed] = [[(h["tokens"].tolist(), round(float(h["score"]), 3)) for h in hb] for hb in hyps]
            assert out[False] == out[True], ctc_w
    finally:
        K.ffn_configure(split=old_split)


def test_config5a_greedy_ids_equal_the_padded_layout():
    """BASELINE.json configuration 5a at its literal size (12-layer Conformer + CTC head, 256 x 1000 x 80, bf16, V = 10 000):
    the greedy token ids of the packed layout are those of the padded one, bit for bit (64 000 rows: one workgroup per row
    block in every row-wise kernel, so no summation order depends on where a frame sits)."""
    import bench

    torch.manual_seed(3)
    a = M.recipe_args(conformer=True, vocab_size=10000, ctc_weight=1.0)
    model = M.S2TCTCModel.build_model(a, M.FakeTask(10000)).prepare(torch.bfloat16, DEV)
    model.encoder.ctc_out_dtype = torch.float32
    model.eval()
    sample, _ = bench.synthetic_batch(256, 1000, 10000, 2, DEV)
    ids, sc = {}, {}
    with torch.no_grad():
        for packed in (False, True):
            with _layout(packed):
                hyp = M.CTCDecoder([model]).generate([model], sample)
                ids[packed] = [h[0]["tokens"].tolist() for h in hyp]
                sc[packed] = torch.stack([h[0]["score"].reshape(()) for h in hyp])
        with _layout(True):  # packing on, but the decoder asked for the reference's scores: it takes the padded layout
            dec = M.CTCDecoder([model])
            dec.exact_scores = True
            hyp = dec.generate([model], sample)
            ids["exact"] = [h[0]["tokens"].tolist() for h in hyp]
            sc["exact"] = torch.stack([h[0]["score"].reshape(()) for h in hyp])
    assert ids[False] == ids[True] == ids["exact"]
    assert sum(len(x) for x in ids[True]) > 1000
    # hypothesis scores (INTEGRATION.md "Hypothesis scores"): the padded layout sums the padded frames' non-blank top-1
    # log-probabilities as the reference does (s2t_ctc.py:327-329); the packed one cannot.  exact_scores restores them bit for bit;
    # the full-length utterance (no padded frame) has the same score either way up to the summation order of its frames.
    assert torch.equal(sc["exact"], sc[False])
    lens = sample["net_input"]["src_lengths"].cpu()
    full = (lens == lens.max()).nonzero().flatten().tolist()
    for b in full:
        assert abs(float(sc[True][b]) - float(sc[False][b])) <= 1e-4 * max(1.0, abs(float(sc[False][b])))
    assert bool((sc[True] <= sc[False] + 1e-3).all())  # the missing term is a sum of -log p >= 0 (scores are -sum lprob)


def _eval_both(model, sample, ni):
    outs = {}
    with torch.no_grad():
        for packed in (False, True):
            with _layout(packed):
                enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
                assert (enc.get("packed") is not None) == packed
                logits, _ = model.decoder(prev_output_tokens=ni["prev_output_tokens"], encoder_out=enc)
                if ni["prev_output_tokens"].numel() >= 2048:  # the target rows run packed as well (what the criterion asks for)
                    lp, ex = model.decoder(prev_output_tokens=ni["prev_output_tokens"], encoder_out=enc, packed_out=True)
                    assert ("packed" in ex) == packed and (lp.dim() == 2) == packed
                hyp = M.CTCDecoder([model.encoder]).generate([model.encoder], sample)
                outs[packed] = (enc["encoder_out"][0].float(), enc["ctc_logit"][0].float(), logits.float(),
                                [h[0]["tokens"].tolist() for h in hyp])
    return outs


@pytest.mark.parametrize("conformer", [True, False])
def test_training_step_equals_the_padded_layout(conformer):
    """Loss, every parameter gradient and the BatchNorm running statistics of one training pass (no dropout: its masks are
    indexed by row, which the layouts number differently)."""
    res = {}
    sample, _ = _sample(24, 1000, 5)
    for packed in (False, True):
        model = _model(conformer)
        model.train()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        with _layout(packed):
            model.flat.zero_grad()
            loss, _, log = crit(model, sample)
            loss.backward()
            torch.cuda.synchronize()
        grads = {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}
        bufs = {k: b.detach().float().clone() for k, b in model.named_buffers() if "running" in k}
        res[packed] = (float(loss.detach()), float(log["ctc_loss"]), grads, bufs)
    la, ca, ga, ba = res[False]
    lb, cb, gb, bb = res[True]
    assert abs(la - lb) <= 2e-4 * abs(la) and abs(ca - cb) <= 2e-4 * abs(ca), (la, lb, ca, cb)
    for k in ba:
        assert torch.allclose(ba[k], bb[k], rtol=2e-4, atol=2e-5), k  # (fp32 sums of bf16 values in another order)
    errs = []
    for k in ga:
        den = float(ga[k].norm())
        err = float((ga[k] - gb[k]).norm()) / max(den, 1e-6)
        if k.endswith(("k_proj.bias", "linear_k.bias")) or den < 1e-5:  # mathematically zero (softmax ignores a key bias)
            continue
        errs.append(err)
        # bf16 activations under another summation order of the BatchNorm statistics and of the fused feed-forward partial
        # rows (tools/grad_noise.py: the same model under two summation orders of ONE layout moves single tensors by as much)
        assert err <= 0.08, (k, err)
    assert float(np.median(errs)) <= 0.02, float(np.median(errs))


@pytest.mark.parametrize("conformer", [True, False])
def test_pds_stages_equal_the_padded_layout(conformer):
    """Progressive down-sampling (models/speech_to_text/pdss2t_transformer.py:1042-1281; 3 stages, d = 256): a stage's layers run
    on packed rows between its down-sampling convolution (padded rows) and the next one — every stage of the Transformer form,
    the stages of at most 256 frames of the Conformer form (the relative-position backward's limit); one training pass against
    the padded layout (loss, gradients, BatchNorm statistics) and the eval logits."""
    from s2t_amd import pdss2t_transformer as PDS

    def build():
        torch.manual_seed(9)
        a = M.recipe_args(conformer=conformer, vocab_size=V, arch="pdss2t_transformer_s_8", pds_stages=3, pds_layers="1_1_1",
                          pds_ratios="2_2_2", pds_fusion=False, pds_embed_dims="256_256_256", pds_ds_method="conv",
                          pds_embed_norm=True, pds_position_embed="1_1_1", pds_kernel_sizes="5_5_5", pds_ffn_ratios="8_8_8",
                          pds_attn_heads="4_4_4", decoder_layers=1)
        m = PDS.PDSS2TTransformerModel.build_model(a, M.FakeTask(V))
        g = torch.Generator().manual_seed(10)
        with torch.no_grad():
            for n_, p in m.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
        return m.prepare(torch.bfloat16, DEV)

    sample, lens = _sample(32, 1200, 21)
    ni = sample["net_input"]
    res = {}
    for packed in (False, True):
        model = build()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        with _layout(packed):
            model.eval()
            with torch.no_grad():
                enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
                assert (enc.get("packed") is not None) == packed
                logit = enc["ctc_logit"][0].float()
            model.train()
            model.flat.zero_grad()
            loss, _, log = crit(model, sample)
            loss.backward()
            torch.cuda.synchronize()
        res[packed] = (logit, float(loss.detach()), {k: p.grad.detach().float().clone() for k, p in model.named_parameters()},
                       {k: b.detach().float().clone() for k, b in model.named_buffers() if "running" in k})
    Tp = res[False][0].shape[0]
    sub = torch.tensor(lens)
    for r in (2, 2, 2):
        sub = torch.floor((sub.float() - 1) / r + 1).long()
    valid = (torch.arange(Tp)[:, None] < sub[None, :]).to(DEV)
    a, b = res[False][0][valid], res[True][0][valid]
    assert float((a - b).norm() / a.norm()) <= 5e-3
    assert abs(res[False][1] - res[True][1]) <= 5e-4 * abs(res[False][1]), (res[False][1], res[True][1])
    for k in res[False][3]:
        assert torch.allclose(res[False][3][k], res[True][3][k], rtol=2e-4, atol=2e-5), k
    errs = []
    for k, ga in res[False][2].items():
        den = float(ga.norm())
        if k.endswith(("k_proj.bias", "linear_k.bias")) or den < 1e-5:
            continue
        e = float((ga - res[True][2][k]).norm()) / den
        errs.append(e)
        assert e <= 0.08, (k, e)
    assert float(np.median(errs)) <= 0.02


def test_one_captured_step_serves_batches_of_any_fill():
    """The packed step captured into a hipGraph and replayed on batches of different fill gives what eager packed steps on
    the same sequence give (the live row count is read on the device: nothing in the graph depends on a batch's lengths),
    and both start from the padded layout's losses."""
    traj = {}
    batches = [_sample(24, 1000, 11 + i, full_first=(i % 2 == 0), lo=0.5 + 0.1 * i)[0] for i in range(3)]
    seq = [0, 1, 2, 2, 2, 0, 1, 2]  # capture() runs two eager updates on its batch before recording the step
    for mode in ("padded_eager", "packed_eager", "packed_graph"):
        model = _model(True, dropout=0.0)
        model.train()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        with _layout(mode != "padded_eager"):
            tr = TR.Trainer(model, crit, lr=1e-5, warmup_updates=1, clip_norm=10.0)  # (small steps: the losses follow the batches)
            losses = {}
            if mode != "packed_graph":
                for i, bi in enumerate(seq):
                    losses[i] = float(tr.train_step(batches[bi])[0])
            else:
                for i in (0, 1):
                    losses[i] = float(tr.train_step(batches[seq[i]])[0])
                # (the captured batch object becomes the graph's static batch, which later replays overwrite: a copy of its own)
                static = {"net_input": {k: v.clone() for k, v in batches[2]["net_input"].items()},
                          "target": batches[2]["target"].clone(), "ntokens": batches[2]["ntokens"]}
                tr.capture(static)  # updates 2 and 3
                losses[4] = float(tr.replay()[0])
                for i in (5, 6, 7):
                    losses[i] = float(tr.replay(batches[seq[i]])[0])
            torch.cuda.synchronize()
        traj[mode] = losses
    pad, eag, gra = traj["padded_eager"], traj["packed_eager"], traj["packed_graph"]
    assert abs(pad[0] - pad[1]) > 1e-3 * abs(pad[0])  # the batches really differ
    for i in sorted(gra):
        assert np.isfinite(gra[i]) and abs(eag[i] - gra[i]) <= 2e-3 * abs(eag[i]), (i, eag, gra)
    for i in sorted(pad):
        assert abs(pad[i] - eag[i]) <= (1e-4 if i == 0 else 5e-3) * abs(pad[i]), (i, pad, eag)


def test_captured_steps_survive_eager_passes_and_a_second_trainer():
    """The per-batch bookkeeping a captured step reads (functional.batch_memo: lengths, masks, positions, packed-row geometry)
    lives OUTSIDE its hipGraph, which bakes the addresses in.  Round 4 shelved a memory fault of a process that built, dropped
    and rebuilt Trainers; the audit behind it ("Per-batch bookkeeping") found the table keyed by ``id(owner)`` —
    re-used by CPython once the owner is collected — and an eager pass over ANOTHER batch object replacing (and releasing) the
    entries of the captured batch.  Now: owners carry process-unique serials, the captured batch's entries are pinned by
    Trainer.capture and refreshed in place.  Checked here:
      * trainer 1: capture, replay, an eager eval pass AND an eager training-mode forward over another batch object on the same
        model, replays on new batches — the losses equal those of a twin that never ran the eager passes;
      * trainer 1 dropped (graphs destroyed, cache emptied), trainer 2 on a NEW model of the same architecture (the collected
        modules' ids are up for re-use): capture and replays follow its own eager twin;
      * nothing of trainer 1 is left pinned."""
    import gc

    batches = [_sample(24, 1000, 31 + i, full_first=(i % 2 == 0), lo=0.5 + 0.1 * i)[0] for i in range(4)]
    other = _sample(24, 1000, 77, full_first=False, lo=0.7)[0]

    def clone(bt):
        return {"net_input": {k: v.clone() for k, v in bt["net_input"].items()}, "target": bt["target"].clone(),
                "ntokens": bt["ntokens"]}

    def run(seed, disturb):
        model = _model(True, seed=seed, dropout=0.0)
        model.train()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        tr = TR.Trainer(model, crit, lr=1e-5, warmup_updates=1, clip_norm=10.0)
        out = []
        tr.capture(clone(batches[0]))
        out.append(float(tr.replay()[0]))
        if disturb:
            model.eval()
            with torch.no_grad():
                enc = model.encoder(src_tokens=other["net_input"]["src_tokens"], src_lengths=other["net_input"]["src_lengths"])
                model.decoder(prev_output_tokens=other["net_input"]["prev_output_tokens"], encoder_out=enc)
            model.train()
            with torch.no_grad():
                crit(model, other)
        for bt in batches[1:]:
            out.append(float(tr.replay(bt)[0]))
        out.append(float(tr.replay(batches[0])[0]))
        torch.cuda.synchronize()
        return out, tr, model

    with _layout(True):
        a, tr_a, m_a = run(0, disturb=True)
        assert any(len(e) > 4 and e[4] == Fn.memo_owner(tr_a) for lst in Fn._MEMO_PINNED.values() for e in lst)  # something IS pinned
        owner_a = Fn.memo_owner(tr_a)
        tr_a.release()
        del tr_a, m_a
        gc.collect()
        torch.cuda.empty_cache()
        assert not any(len(e) > 4 and e[4] == owner_a for lst in Fn._MEMO_PINNED.values() for e in lst)
        b, tr_b, m_b = run(0, disturb=False)
        tr_b.release()
        del tr_b, m_b
        gc.collect()
        torch.cuda.empty_cache()
        c, tr_c, m_c = run(5, disturb=True)   # a second architecture-equal model behind two dropped ones
        d, tr_d, m_d = run(5, disturb=False)  # ... and two captured trainers alive at once
        torch.cuda.synchronize()
    assert all(np.isfinite(a + b + c + d))
    for x, y in ((a, b), (c, d)):
        for i, (u, v) in enumerate(zip(x, y)):
            assert abs(u - v) <= 2e-4 * abs(v), (i, x, y)  # (identical kernels and data: float atomics in parameter sums only)
    assert abs(a[0] - a[1]) > 1e-3 * abs(a[0])  # the batches really differ


def test_sate_textual_layers_run_packed_and_equal_the_padded_layout():
    """Configuration 4's stack (s2t_sate: acoustic Transformer encoder -> inter_league adapter -> textual layers -> decoder,
    models/speech_to_text/s2t_sate.py:973-1075) with the acoustic rows handed to the adapter and the textual layers AS THEY ARE:
    eval outputs on the frames equal the padded layout's bit for bit (one workgroup per fused-FFN row block pinned), one training
    pass agrees in loss and in every gradient within the spread of two summation orders."""
    from s2t_amd import s2t_sate as SATE

    def build():
        torch.manual_seed(3)
        a = M.recipe_args(conformer=False, vocab_size=V, arch="s2t_sate", encoder_layers=3, text_encoder_layers=2, decoder_layers=2,
                          acoustic_encoder="transformer", adapter="inter_league", textual_encoder_embed_norm=True,
                          textual_encoder_no_scale_embedding=True, encoder_normalize_before=True, decoder_normalize_before=True)
        model = SATE.S2TSATEModel.build_model(a, M.FakeTask(V))
        g = torch.Generator().manual_seed(10)
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
        return model.prepare(torch.bfloat16, DEV)

    sample, lens = _sample(24, 1000, 15)
    ni = sample["net_input"]
    sub = torch.tensor(lens)
    for _ in range(2):
        sub = torch.div(sub - 1, 2, rounding_mode="floor") + 1
    _, old, _ = K.ffn_configure()
    res = {}
    for packed in (False, True):
        model = build()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        with _layout(packed):
            K.ffn_configure(split=1)
            try:
                model.eval()
                with torch.no_grad():
                    enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
                    assert (enc.get("packed") is not None) == packed
                    eo = enc["encoder_out"][0].float()
                    logits, _ = model.decoder(prev_output_tokens=ni["prev_output_tokens"], encoder_out=enc)
            finally:
                K.ffn_configure(split=old)
            model.train()
            model.flat.zero_grad()
            loss, _, log = crit(model, sample)
            loss.backward()
            torch.cuda.synchronize()
        res[packed] = (eo, logits.float(), float(loss.detach()), {k: p.grad.detach().float().clone() for k, p in model.named_parameters()})
    Tp = res[False][0].shape[0]
    valid = (torch.arange(Tp)[:, None] < sub[None, :]).to(DEV)
    assert torch.equal(res[False][0][valid], res[True][0][valid])          # textual encoder output on the frames
    tmask = ni["prev_output_tokens"].ne(1)
    assert torch.equal(res[False][1][tmask], res[True][1][tmask])          # decoder logits on the target positions
    assert abs(res[False][2] - res[True][2]) <= 5e-4 * abs(res[False][2]), (res[False][2], res[True][2])
    errs = []
    for k, ga in res[False][3].items():
        den = float(ga.norm())
        if k.endswith(("k_proj.bias", "linear_k.bias")) or den < 1e-5:
            continue
        e = float((ga - res[True][3][k]).norm()) / den
        errs.append(e)
        assert e <= 0.08, (k, e)
    assert float(np.median(errs)) <= 0.02


@pytest.mark.parametrize("conformer", [True, False])
def test_intermediate_ctc_taps_run_packed_and_equal_the_padded_layout(conformer):
    """Intermediate CTC heads with prediction-aware encoding (egs/mustc/asr/conf/inter.yaml; models/speech_to_text/
    s2t_transformer.py:1881-1946: tap LayerNorm -> shared head -> x = PAE(norm_x, logits)) on packed rows: the taps' LayerNorm,
    projection, softmax and embedding product run on the frames only, the criterion takes the packed logits of every head.
    Eval: encoder output and every head's logits on the frames equal the padded layout's bit for bit; one training pass: the
    joint loss with the intermediate term and every gradient within the spread of two summation orders."""
    def build():
        torch.manual_seed(4)
        a = M.recipe_args(conformer=conformer, vocab_size=V, encoder_layers=4, decoder_layers=1, inter_ctc_layers="2,3",
                          share_inter_ctc=True, inter_ctc_weight=0.2, ctc_pae="inter_league")
        model = M.S2TTransformerModel.build_model(a, M.FakeTask(V))
        g = torch.Generator().manual_seed(12)
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
            for n_, b in model.named_buffers():
                if n_.endswith("running_mean"):
                    b.copy_(0.1 * torch.randn(b.shape, generator=g))
                if n_.endswith("running_var"):
                    b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
        return model.prepare(torch.bfloat16, DEV)

    sample, lens = _sample(24, 1000, 25)
    ni = sample["net_input"]
    sub = torch.tensor(lens)
    for _ in range(2):
        sub = torch.div(sub - 1, 2, rounding_mode="floor") + 1
    _, old, _ = K.ffn_configure()
    res = {}
    for packed in (False, True):
        model = build()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3, inter_ctc_weight=0.2)
        with _layout(packed):
            K.ffn_configure(split=1)
            try:
                model.eval()
                with torch.no_grad():
                    enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
                    assert (enc.get("packed") is not None) == packed
                    assert len(enc["inter_ctc_logits"]) == 2
                    outs = [enc["encoder_out"][0].float(), enc["ctc_logit"][0].float()] + [il[0].float() for il in enc["inter_ctc_logits"]]
                    assert all(torch.equal(il[1], enc["encoder_padding_mask"][0]) for il in enc["inter_ctc_logits"])
            finally:
                K.ffn_configure(split=old)
            model.train()
            model.flat.zero_grad()
            loss, _, log = crit(model, sample)
            loss.backward()
            torch.cuda.synchronize()
        res[packed] = (outs, float(loss.detach()), float(log["inter_ctc_loss"]),
                       {k: p.grad.detach().float().clone() for k, p in model.named_parameters()})
    Tp = res[False][0][0].shape[0]
    valid = (torch.arange(Tp)[:, None] < sub[None, :]).to(DEV)
    for a_, b_ in zip(res[False][0], res[True][0]):
        assert torch.equal(a_[valid], b_[valid])
    assert abs(res[False][1] - res[True][1]) <= 5e-4 * abs(res[False][1]), (res[False][1], res[True][1])
    assert abs(res[False][2] - res[True][2]) <= 5e-4 * abs(res[False][2]), (res[False][2], res[True][2])
    errs = []
    for k, ga in res[False][3].items():
        den = float(ga.norm())
        if k.endswith(("k_proj.bias", "linear_k.bias")) or den < 1e-5:
            continue
        e = float((ga - res[True][3][k]).norm()) / den
        errs.append(e)
        assert e <= 0.08, (k, e)
    assert float(np.median(errs)) <= 0.02


def test_nast_stack_at_d512_decodes_on_packed_rows_in_inference():
    """Configuration 5b's stack at the recipe's width (egs/mustc/st/conf/reproduction_nast.yaml: d = 512, 8 heads of 64,
    F = 2048): Conformer acoustic layers with shared intermediate CTC heads + prediction-aware encoding, the inter_league adapter,
    textual layers with the cross-layer attention (modules/transformer_s2_layer.py:214-336), intermediate XCTC heads + PAE and
    the XCTC head (s2t_sate.py:692-808) — in INFERENCE the whole stack runs on the frames only (the LayerNorm / GEMM composition
    takes the row map at any width; training at this width stays padded).  Every logit family equals the padded layout's on the
    frames bit for bit, and greedy CTC decoding (s2t_ctc.py:174-349) takes the packed XCTC rows as they are: same token ids."""
    Vn = 2000
    nast = dict(encoder_type="sate", text_encoder_layers=3, acoustic_encoder="transformer", adapter="inter_league",
                xctc_weight=1.0, ctc_weight=1.0, share_ctc_and_embed=True, share_xctc_and_embed=True, text_no_pos_emb=True,
                textual_encoder_embed_norm=False, textual_encoder_no_scale_embedding=True, encoder_normalize_before=True,
                share_inter_ctc=True, inter_ctc_weight=1.0, inter_ctc_layers="1", inter_xctc_weight=1.0, inter_xctc_layers="2",
                ctc_pae="inter_league", xctc_pae="inter_league", xctc_cross_attn=True, cross_attn_start_layer=2,
                cross_attn_layer=1, cross_attn_collaboration_mode="serial", cross_attn_league_drop_net=True,
                cross_attn_league_drop_net_prob=0.1, xctc_pae_ground_truth_ratio=0.8, xctc_pae_ground_truth_only_mistake=True,
                pae_oracle_smooth=True, encoder_embed_dim=512, encoder_ffn_embed_dim=2048, encoder_attention_heads=8,
                encoder_layers=2, subsampling_filter=2048, activation_fn="relu", arch="s2t_ctc")
    a = M.recipe_args(conformer=True, vocab_size=Vn, **nast)
    torch.manual_seed(8)
    model = M.S2TCTCModel.build_model(a, M.FakeTask(Vn))
    g = torch.Generator().manual_seed(18)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for n_, b in model.named_buffers():
            if n_.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            if n_.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
    model.prepare(torch.bfloat16, DEV)
    model.eval()
    model.encoder.xctc_out_dtype = torch.float32  # the decoded head in fp32 (bit-exact arg-max)
    sample, lens = _sample(24, 1000, 31)
    ni = sample["net_input"]
    sub = torch.tensor(lens)
    for _ in range(2):
        sub = torch.div(sub - 1, 2, rounding_mode="floor") + 1
    dec = M.CTCDecoder([model], None, None, blank_idx=0)
    res = {}
    for packed in (False, True):
        with _layout(packed), torch.no_grad():
            enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
            pk = enc.get("packed")
            assert (pk is not None) == packed
            if packed:  # every family rides as packed rows; nothing was unpacked to get here
                assert pk["xctc_logit"] is not None and pk["xctc_logit"].dtype == torch.float32
                assert len(pk["inter_ctc_logit"]) == 1 and len(pk["inter_xctc_logit"]) == 1
            fam = {"encoder_out": enc["encoder_out"][0], "ctc": enc["ctc_logit"][0], "xctc": enc["xctc_logit"][0],
                   "inter_ctc": enc["inter_ctc_logits"][0], "inter_xctc": enc["inter_xctc_logits"][0]}
            fam = {k: (v[0] if isinstance(v, (list, tuple)) else v).float() for k, v in fam.items()}
            hyps = dec.generate([model], sample)
        res[packed] = (fam, [h[0]["tokens"].cpu() for h in hyps])
    Tp = res[False][0]["encoder_out"].shape[0]
    valid = (torch.arange(Tp)[:, None] < sub[None, :]).to(DEV)
    for k in res[False][0]:
        a_, b_ = res[False][0][k], res[True][0][k]
        assert a_.shape == b_.shape, k
        assert torch.equal(a_[valid], b_[valid]), (k, float((a_[valid] - b_[valid]).abs().max()))
    assert sum(len(t) for t in res[False][1]) > 0
    for ta, tb in zip(res[False][1], res[True][1]):
        assert torch.equal(ta, tb)
    # training at this width keeps the padded layout (the packed weight gradients exist on the d = 256 kernels only)
    model.train()
    enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
    assert enc.get("packed") is None


@pytest.mark.parametrize("conformer", [True, False])
def test_an_eager_pass_over_a_batch_overwritten_in_place_sees_the_new_batch(conformer):
    """a static batch whose tensors are overwritten IN PLACE (``copy_`` of a new batch into the same tensors,
    without Trainer.load_batch) and then run eagerly.  The memos made from the raw tensors notice the moved version counters; the
    memos made from THEIR outputs (CTC input lengths and key masks from the padding mask, the packed row map from the int32
    lengths, the packed targets) are written through raw addresses by the one-launch forms, so their own counters never move —
    the refresh must cascade.  Loss, CTC-greedy ids and every gradient of the overwritten batch equal those of the same data in
    fresh tensors."""
    model = _model(conformer)
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    first, _ = _sample(24, 1000, 41)
    second, lens2 = _sample(24, 1000, 42, lo=0.3)  # other lengths (shorter utterances), other targets, the same shapes
    assert not torch.equal(first["net_input"]["src_lengths"], second["net_input"]["src_lengths"])

    def evaluate(sample):
        model.eval()
        with torch.no_grad():
            enc = model.encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
            logits, _ = model.decoder(sample["net_input"]["prev_output_tokens"], encoder_out=enc)
            return enc["ctc_logit"][0].float().argmax(-1).clone(), enc["encoder_padding_mask"][0].clone(), logits.float().clone()

    def train(sample):
        model.train()
        model.flat.zero_grad()
        loss, _, _ = crit(model, sample)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), model.flat.grad.clone()

    static = {"net_input": {k: v.clone() for k, v in first["net_input"].items()}, "target": first["target"].clone(),
              "ntokens": first["ntokens"]}
    # eval passes first (a training pass moves the BatchNorm running statistics the eval passes read)
    ev_first = evaluate(static)
    for k, v in second["net_input"].items():   # the new batch INTO the same tensor objects
        static["net_input"][k].copy_(v)
    static["target"].copy_(second["target"])
    static["ntokens"] = second["ntokens"]
    ev_got = evaluate(static)
    ev_want = evaluate(second)
    assert not torch.equal(ev_first[1], ev_want[1])          # the two batches really differ (other lengths)
    assert torch.equal(ev_got[1], ev_want[1])                # padding mask of the NEW lengths
    assert torch.equal(ev_got[0], ev_want[0])                # CTC-greedy frames
    tm = second["net_input"]["prev_output_tokens"].ne(1)
    assert torch.equal(ev_got[2][tm], ev_want[2][tm])        # decoder logits (packed target rows, key masks of the new lengths)
    # training passes (batch statistics: independent of the running ones): back to the first batch, then the second in place
    for k, v in first["net_input"].items():
        static["net_input"][k].copy_(v)
    static["target"].copy_(first["target"])
    static["ntokens"] = first["ntokens"]
    l_first = train(static)[0]
    for k, v in second["net_input"].items():
        static["net_input"][k].copy_(v)
    static["target"].copy_(second["target"])
    static["ntokens"] = second["ntokens"]
    got = train(static)
    want = train(second)
    assert abs(l_first - want[0]) > 1e-3 * abs(want[0])
    assert abs(got[0] - want[0]) <= 1e-5 * abs(want[0]), (got[0], want[0])
    assert float((got[1] - want[1]).norm() / want[1].norm()) <= 1e-4
