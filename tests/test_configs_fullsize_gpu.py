"""The literal BASELINE.json configurations at sizes the CPU oracle still finishes in seconds, HIP path
against the oracle on the same seeded weights and inputs:

  config 1   s2t_transformer_s 6 enc / 6 dec, d = 256, V = 10000, 32 x 400 x 80, fp32: logits within 1e-3 relative,
             CTC-greedy token ids equal (models/speech_to_text/s2t_ctc.py:236-349)
  config 5a  12-layer Conformer + CTC head, 8 x 1000 x 80 (the 256-utterance batch is 32 of these), fp32
  config 3   PDS Conformer (4 stages, ratios 2-2-1-2, d = 256) at 4 x 2000 x 80: stage lengths 1004 / 502 / 502 / 251
             (models/speech_to_text/pdss2t_transformer.py:1042-1281), fp32
  config 2'  d = 256 / 64-wide heads in bf16 (fused attention, fused FFN blocks, ln256 kernels, V = 10000 loss kernels)
             against the oracle run on the bf16-ROUNDED weights: forward and gradient error bounds at twice the measured error

An argmax over 10 000 fp32 logits can legitimately differ between two correct fp32 implementations where the top two
logits tie to rounding: a differing frame is accepted only when the oracle's own top-2 gap there is below 1e-4 of the
logit scale, and at most two such frames per batch.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import s2t_oracle as O  # noqa: E402
from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import pdss2t_transformer as PDS  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402

DEV = "cuda"
V = 10000
CONF = dict(macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, encoder_attention_type="rel_pos",
            encoder_activation_fn="swish", layer_padding_mask=True)


def _batch(B, T, seed, U=None):
    g = torch.Generator().manual_seed(seed)
    lens = sorted([T] + [int(torch.randint(int(0.6 * T), T + 1, (1,), generator=g)) for _ in range(B - 1)], reverse=True)
    src = torch.randn(B, T, 80, generator=g)
    for b, l in enumerate(lens):
        src[b, l:] = 0
    return src, torch.tensor(lens), g


def _targets(B, g, lo, hi):
    """Left-aligned targets of lo .. hi-1 tokens + eos, pad = 1; ``prev`` = eos-shifted (the collater's layout)."""
    ul = [int(torch.randint(lo, hi, (1,), generator=g)) for _ in range(B)]
    U = max(ul) + 1
    target = torch.full((B, U), 1, dtype=torch.long)
    prev = torch.full((B, U), 1, dtype=torch.long)
    for b, u in enumerate(ul):
        toks = torch.randint(4, V, (u,), generator=g)
        target[b, :u] = toks
        target[b, u] = 2
        prev[b, 0] = 2
        prev[b, 1:u + 1] = toks
    return target, prev, ul


class _PackingSpy:
    """Records which users (encoder / decoder) attached a packed-row geometry while active (s2t_amd/rows.py)."""

    def __enter__(self):
        from s2t_amd import rows as Rows

        self.Rows, self.orig, self.tags = Rows, Rows.attach, []

        def attach(lens32, B, T, halo, tag=None):
            self.tags.append(tag[0] if isinstance(tag, tuple) else tag)
            out = self.orig(lens32, B, T, halo, tag)
            assert Rows.K.rows_geom(out) is not None
            return out

        Rows.attach = attach
        return self

    def __exit__(self, *exc):
        self.Rows.attach = self.orig
        return False


# The batch of the bf16 oracle comparisons: 18 x 1000 frames -> 4 500 encoder rows and 18 x 121 = 2 178 target rows, above BOTH
# packing thresholds of s2t_amd/rows.py at their defaults (4096 / 2048): the layout, kernels and launch geometry of the bench.
PK_B, PK_T, PK_ULO, PK_UHI = 18, 1000, 90, 121


def _perturb(model, seed):
    """Nothing left at a trivial initial value: LayerNorm / BatchNorm gains, biases and running statistics."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for n_, b in model.named_buffers():
            if n_.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            if n_.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))


def _rel(got, ref):
    got = got.detach().float().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))


def _greedy_equal_up_to_ties(model_enc, src, lens, logit_o, mask_o):
    hy, _ = O.ctc_greedy(logit_o, mask_o)
    dec = M.CTCDecoder([model_enc], None, None)

    class _Enc(torch.nn.Module):
        def __init__(self, e):
            super().__init__()
            self.e = e

        def forward(self, src_tokens, src_lengths):
            return self.e(src_tokens, src_lengths)

    dec.model = _Enc(model_enc)
    hyps = dec.generate(None, {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV)}})
    got = [h[0]["tokens"].tolist() for h in hyps]
    ref = [h.tolist() for h in hy]
    bad = [b for b in range(len(ref)) if got[b] != ref[b]]
    if bad:
        top2 = logit_o.float().topk(2, dim=-1).values  # (T', B, 2)
        gap = (top2[..., 0] - top2[..., 1]) / logit_o.float().abs().amax(-1).clamp_min(1e-6)
        assert len(bad) <= 2, "CTC-greedy ids differ in %d utterances" % len(bad)
        for b in bad:
            assert float(gap[:, b].min()) < 1e-4, "utterance %d differs without a top-2 tie (min gap %.3g)" % (b, float(gap[:, b].min()))
    return len(bad)


def test_config1_transformer_6_6_fp32_logits_and_greedy_ids():
    torch.manual_seed(21)
    args = M.recipe_args(conformer=False, vocab_size=V, encoder_layers=6, decoder_layers=6)
    model = M.S2TTransformerModel.build_model(args, M.FakeTask(V))
    _perturb(model, 22)
    W = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    model.prepare(torch.float32, DEV)
    model.eval()
    src, lens, g = _batch(32, 400, 23)
    prev = torch.randint(4, V, (32, 12), generator=g)
    prev[:, 0] = 2
    with torch.no_grad():
        enc = model.encoder(src.to(DEV), lens.to(DEV))
        logits, _ = model.decoder(prev.to(DEV), encoder_out=enc)
        enc_o = O.encoder_forward(src, lens, W, cfg, training=False)
        logits_o = O.decoder_forward(prev, enc_o, W, cfg)
    assert _rel(enc["encoder_out"][0], enc_o["encoder_out"][0]) < 1e-3
    assert _rel(enc["ctc_logit"][0], enc_o["ctc_logit"][0]) < 1e-3
    assert _rel(logits, logits_o) < 1e-3
    model.encoder.ctc_out_dtype = torch.float32
    _greedy_equal_up_to_ties(model.encoder, src, lens, enc_o["ctc_logit"][0], enc_o["encoder_padding_mask"][0])


def test_config5a_conformer12_ctc_fp32_8x1000():
    torch.manual_seed(31)
    args = M.recipe_args(conformer=True, vocab_size=V, ctc_weight=1.0)
    model = M.S2TCTCModel.build_model(args, M.FakeTask(V))
    _perturb(model, 32)
    W = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    model.prepare(torch.float32, DEV)
    model.eval()
    src, lens, g = _batch(8, 1000, 33)
    with torch.no_grad():
        enc = model.encoder(src.to(DEV), lens.to(DEV))
        enc_o = O.encoder_forward(src, lens, W, cfg, training=False)
    assert _rel(enc["encoder_out"][0], enc_o["encoder_out"][0]) < 1e-3
    assert _rel(enc["ctc_logit"][0], enc_o["ctc_logit"][0]) < 1e-3
    model.encoder.ctc_out_dtype = torch.float32
    _greedy_equal_up_to_ties(model.encoder, src, lens, enc_o["ctc_logit"][0], enc_o["encoder_padding_mask"][0])


def _pds_args(**kw):
    return M.recipe_args(conformer=True, vocab_size=V, arch="pdss2t_transformer_s_8", pds_stages=4, pds_layers="3_3_3_3",
                         pds_ratios="2_2_1_2", pds_fusion=False, pds_embed_dims="256_256_256_256", pds_ds_method="conv",
                         pds_embed_norm=True, pds_position_embed="1_1_1_1", pds_kernel_sizes="5_5_5_5",
                         pds_ffn_ratios="8_8_8_8", pds_attn_heads="4_4_4_4", **kw)


def test_config3_pds_conformer_fp32_4x2000():
    """The recipe's 4-stage PDS Conformer at the configuration's 2000-frame length (3 layers per stage, d = 256): the
    relative-position attention runs at T' = 1004 and 502 — through the GEMM-composed fp32 path here, through the fused
    bf16 kernels in test_attn_fused_gpu.py at the same lengths."""
    torch.manual_seed(41)
    args = _pds_args()
    model = PDS.PDSS2TTransformerModel.build_model(args, M.FakeTask(V))
    _perturb(model, 42)
    W = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    model.prepare(torch.float32, DEV)
    model.eval()
    src, lens, g = _batch(4, 2000, 43)
    prev = torch.randint(4, V, (4, 9), generator=g)
    prev[:, 0] = 2
    with torch.no_grad():
        enc = model.encoder(src.to(DEV), lens.to(DEV))
        logits, _ = model.decoder(prev.to(DEV), encoder_out=enc)
        enc_o = O.pds_encoder_forward(src, lens, W, cfg, training=False)
        logits_o = O.decoder_forward(prev, enc_o, W, cfg)
    assert enc["encoder_out"][0].shape[0] == 251
    assert _rel(enc["encoder_out"][0], enc_o["encoder_out"][0]) < 1e-3
    assert _rel(enc["ctc_logit"][0], enc_o["ctc_logit"][0]) < 1e-3
    assert _rel(logits, logits_o) < 1e-3


def test_config2p_bf16_d256_against_oracle_on_rounded_weights():
    """The kernels of the headline step IN THE LAYOUT THE BENCH RUNS — packed encoder rows and packed target rows
    (s2t_amd/rows.py), fused rel-pos attention with 64-wide heads, the 128-row fused FFN blocks, 256-wide LayerNorm kernels,
    the grouped weight gradients over the live row count, the V = 10000 loss kernels — inside a 4-layer d = 256 Conformer at
    18 x 1000 with 90 - 120 target tokens (4 500 encoder rows, 2 178 target rows: both above the packing thresholds at their
    defaults, asserted below), in bf16, against the fp32 oracle evaluated on the SAME bf16-rounded weights and inputs: what is
    left is activation rounding only."""
    from s2t_amd import rows as Rows

    assert Rows.ENABLED and PK_B * 250 >= Rows.MIN_ENC_ROWS and PK_B * PK_UHI >= Rows.MIN_DEC_ROWS
    torch.manual_seed(51)
    args = M.recipe_args(conformer=True, vocab_size=V, encoder_layers=4, decoder_layers=2)
    model = M.S2TTransformerModel.build_model(args, M.FakeTask(V))
    _perturb(model, 52)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.bfloat16().float())
    W = {k: v.detach().clone().float().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    model.prepare(torch.bfloat16, DEV)
    model.train()
    B, T = PK_B, PK_T
    src, lens, g = _batch(B, T, 53)
    src = src.bfloat16().float()
    target, prev, ul = _targets(B, g, PK_ULO, PK_UHI)
    assert B * target.shape[1] >= Rows.MIN_DEC_ROWS
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    sample = {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": int(sum(ul) + B)}
    with _PackingSpy() as spy:
        model.flat.zero_grad()
        loss, _, log = crit(model, sample)
        loss.backward()
        torch.cuda.synchronize()
    assert "enc" in spy.tags and "dec" in spy.tags, spy.tags  # both sides ran packed
    loss_o, aux = O.joint_loss(W, cfg, src, lens, prev, target, eps=0.1, training=True, use_torch_ctc=True)
    loss_o.backward()
    lo = float(loss_o.detach())
    assert abs(float(loss.detach()) - lo) < 5e-3 * abs(lo), (float(loss.detach()), lo)
    ptr = {k: v.data_ptr() for k, v in model.state_dict().items()}
    errs = {}
    for k, p in model.named_parameters():
        if k.endswith(("k_proj.bias", "linear_k.bias")):
            continue  # mathematically zero
        go = sum(W[k2].grad for k2 in W if ptr[k2] == ptr[k] and W[k2].grad is not None)
        if "subsample" in k and go.dim() == 3:
            go = go.permute(0, 2, 1)
        gf = p.grad.detach().float().cpu()
        errs[k] = float((gf - go).norm() / go.norm().clamp_min(1e-6))
    worst = max(errs.items(), key=lambda kv: kv[1])
    print("bf16 d256 gradient relative L2: worst %s %.4f, median %.4f" % (worst[0], worst[1], float(np.median(list(errs.values())))))
    for k_, v_ in sorted(errs.items(), key=lambda kv: -kv[1])[:12]:
        print("    %.4f %s" % (v_, k_))
    # measured on MI355X: worst 0.055 (layer-0 depthwise-conv weight), median 0.008 in round 2.  The figure is CHAOTIC in the
    # summation order of any kernel on the path (round 3, tools/grad_noise.py: the same model and batch through four
    # equally valid variants of the fused feed-forward kernels, three seeds each: medians 0.007 ... 0.019 against an fp32 run
    # of the HIP path, in no consistent order; this test's seed: 0.006 / 0.010 / 0.034 for three of them, worst tensor 0.054 ...
    # 0.100): one-ulp bf16 flips re-seed the rounding noise of everything downstream.  A wrong kernel shows up as an error of
    # order one on some tensor (a transposed weight layout read 1.43), so the bounds sit above the spread, not at twice one draw.
    # Round 5, this batch (18 x 1000, packed rows on both sides): worst 0.0535 (layer-0 linear_pos.weight), median 0.0032.
    assert worst[1] < 1.5e-1, worst
    assert float(np.median(list(errs.values()))) < 2e-2


def test_config2p_bf16_packed_eval_forward_against_oracle_on_rounded_weights():
    """Eval forward of the same model in the packed layout against the oracle on the rounded weights: encoder output, CTC
    logits and decoder logits (each through the reference's T x B x C / B x U x V views, i.e. through Rows.unpack), and
    the CTC-greedy ids of the packed decode against the oracle's on the oracle's own logits up to exact top-2 ties."""
    from s2t_amd import rows as Rows

    torch.manual_seed(151)
    args = M.recipe_args(conformer=True, vocab_size=V, encoder_layers=4, decoder_layers=2)
    model = M.S2TTransformerModel.build_model(args, M.FakeTask(V))
    _perturb(model, 152)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.bfloat16().float())
    W = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    model.prepare(torch.bfloat16, DEV)
    model.eval()
    B, T = PK_B, PK_T
    src, lens, g = _batch(B, T, 153)
    src = src.bfloat16().float()
    target, prev, ul = _targets(B, g, PK_ULO, PK_UHI)
    with torch.no_grad(), _PackingSpy() as spy:
        enc = model.encoder(src.to(DEV), lens.to(DEV))
        logits, _ = model.decoder(prev.to(DEV), encoder_out=enc)
        enc_o = O.encoder_forward(src, lens, W, cfg, training=False)
        logits_o = O.decoder_forward(prev, enc_o, W, cfg)
    assert "enc" in spy.tags and "dec" in spy.tags, spy.tags
    assert enc.get("packed") is not None
    # compared on the FRAMES / target positions: the reference leaves junk in the padded ones (its final LayerNorm and the
    # projections run on them), the packed layout holds none and unpacks them as zero rows
    olen = [((int(l) - 1) // 2 + 1 - 1) // 2 + 1 for l in lens]
    fm = torch.zeros(enc_o["encoder_out"][0].shape[:2], dtype=torch.bool)  # (T', B)
    for b, n in enumerate(olen):
        fm[:n, b] = True
    tm = prev.ne(1)  # (B, U)

    def rel_on(got, ref, m):
        got = got.detach().float().cpu()
        return float((got - ref)[m].abs().max() / ref[m].abs().max().clamp_min(1e-6))

    e1 = rel_on(enc["encoder_out"][0], enc_o["encoder_out"][0], fm)
    e2 = rel_on(enc["ctc_logit"][0], enc_o["ctc_logit"][0], fm)
    e3 = rel_on(logits, logits_o, tm)
    print("packed bf16 eval vs oracle on rounded weights: encoder_out %.4f ctc_logit %.4f decoder logits %.4f" % (e1, e2, e3))
    # bf16 activations through 4 Conformer layers + 2 decoder layers: the padded layout measures 0.6 - 1.2 % of the tensor's
    # largest magnitude (BF16_BOUNDS of test_model_parity_gpu.py are set the same way); per frame the packed layout computes
    # the same arithmetic.  Measured on MI355X (round 5): 0.0111 / 0.0070 / 0.0032.
    assert e1 < 2.5e-2 and e2 < 1.5e-2 and e3 < 1e-2, (e1, e2, e3)
    eo = enc["encoder_out"][0].float().cpu()
    assert float(eo[~fm].abs().max()) == 0.0  # the unpacked view's padded frames are zero rows
    # CTC-greedy ids of the packed decode (bf16 logits) against the oracle's ids: a frame may differ only where the oracle's own
    # top-2 gap is inside the bf16 logit error measured above
    model.encoder.ctc_out_dtype = torch.float32
    hy, _ = O.ctc_greedy(enc_o["ctc_logit"][0], enc_o["encoder_padding_mask"][0])

    class _Enc(torch.nn.Module):
        def __init__(self, e):
            super().__init__()
            self.e = e

        def forward(self, src_tokens, src_lengths):
            return self.e(src_tokens, src_lengths)

    dec = M.CTCDecoder([_Enc(model.encoder)], None, None)
    hyps = dec.generate(None, {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV)}})
    lo = enc_o["ctc_logit"][0].float()
    top2 = lo.topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1]) / lo.abs().amax().clamp_min(1e-6)  # relative to the tensor's scale, like e2
    for b in range(B):
        got, ref = hyps[b][0]["tokens"].tolist(), hy[b].tolist()
        if got != ref:
            assert float(gap[:olen[b], b].min()) < 2 * e2 + 1e-3, (b, float(gap[:olen[b], b].min()), e2)


def _grads_vs_oracle(seed, variants):
    """Relative L2 error of every parameter gradient of the 4-layer d = 256 bf16 Conformer (18 x 1000, 90 - 120 target tokens:
    above both packing thresholds) against the fp32 oracle on the same bf16-rounded weights and inputs, for each kernel variant
    in ``variants`` (name -> context manager factory)."""
    from s2t_amd import functional as Fn

    torch.manual_seed(seed)
    args = M.recipe_args(conformer=True, vocab_size=V, encoder_layers=4, decoder_layers=2)
    ref = M.S2TTransformerModel.build_model(args, M.FakeTask(V))
    _perturb(ref, seed + 1)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.bfloat16().float())
    state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    W = {k: v.detach().clone().float().requires_grad_(v.is_floating_point()) for k, v in state.items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    B, T = PK_B, PK_T
    src, lens, g = _batch(B, T, seed + 2)
    src = src.bfloat16().float()
    target, prev, ul = _targets(B, g, PK_ULO, PK_UHI)
    sample = {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": int(sum(ul) + B)}
    loss_o, _ = O.joint_loss(W, cfg, src, lens, prev, target, eps=0.1, training=True, use_torch_ctc=True)
    loss_o.backward()
    out = {}
    packed = {}
    for name, ctx in variants.items():
        model = M.S2TTransformerModel.build_model(args, M.FakeTask(V))
        model.load_state_dict(state)
        model.prepare(torch.bfloat16, DEV)
        model.train()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        old = Fn._FFN_FUSED_MIN_ROWS
        Fn._FFN_FUSED_MIN_ROWS = 1024
        try:
            with ctx(), _PackingSpy() as spy:
                model.flat.zero_grad()
                loss, _, _ = crit(model, sample)
                loss.backward()
                torch.cuda.synchronize()
        finally:
            Fn._FFN_FUSED_MIN_ROWS = old
        packed[name] = sorted(set(spy.tags))
        assert abs(float(loss.detach()) - float(loss_o.detach())) < 5e-3 * abs(float(loss_o.detach())), name
        ptr = {k: v.data_ptr() for k, v in model.state_dict().items()}
        errs = {}
        for k, p in model.named_parameters():
            if k.endswith(("k_proj.bias", "linear_k.bias")):
                continue  # mathematically zero
            go = sum(W[k2].grad for k2 in W if ptr[k2] == ptr[k] and W[k2].grad is not None)
            if "subsample" in k and go.dim() == 3:
                go = go.permute(0, 2, 1)
            errs[k] = float((p.grad.detach().float().cpu() - go).norm() / go.norm().clamp_min(1e-6))
        out[name] = errs
    out["_packed"] = packed
    return out


def test_config2p_bf16_shipped_kernels_are_as_close_to_the_oracle_as_the_composed_path():
    """The bf16 gradient error of ONE draw is chaotic in the summation order of any kernel on the path (a one-ulp flip re-seeds
    the rounding noise of everything behind it), so a bound on it has to sit far above a draw (the test above).  What
    discriminates: the SAME model, weights and batch (18 x 1000 frames, 2 178 target rows: encoder AND decoder rows packed at the
    default thresholds, asserted) through (a) the kernels the bench runs — packed rows, the 128-row fused feed-forward kernels,
    the fused attention — and (b) the composed path they replaced (padded rows, 64-row feed-forward
    kernels, GEMM-composed attention: scores and probabilities through HBM), each against the fp32 oracle on the rounded weights,
    over three seeds: per tensor the shipped kernels may not be further from the oracle than 2 x the composed path (a wrong
    scale, a dropped bias gradient or a mis-indexed row shows as an error of order one on its tensor against ~0.01 - 0.05), and
    the three-seed means of the worst tensor and of the median stay at the round-2 levels (0.11 / 0.025)."""
    import contextlib
    import os

    from s2t_amd import kernels as K
    from s2t_amd import rows as Rows

    @contextlib.contextmanager
    def shipped():
        yield

    @contextlib.contextmanager
    def composed():
        mask, split, _ = K.ffn_configure()
        old_rows, old_env = Rows.ENABLED, os.environ.get("S2T_ATTN_COMPOSED")
        K.ffn_configure(pc_mask=0)
        Rows.ENABLED = False
        os.environ["S2T_ATTN_COMPOSED"] = "1"
        try:
            yield
        finally:
            K.ffn_configure(pc_mask=mask)
            Rows.ENABLED = old_rows
            if old_env is None:
                del os.environ["S2T_ATTN_COMPOSED"]
            else:
                os.environ["S2T_ATTN_COMPOSED"] = old_env

    seeds = (51, 77, 123)
    runs = [_grads_vs_oracle(s, {"shipped": shipped, "composed": composed}) for s in seeds]
    for r in runs:
        pk = r.pop("_packed")
        assert pk["shipped"] == ["dec", "enc"] and pk["composed"] == [], pk  # (a) ran packed on both sides, (b) padded
    names = list(runs[0]["shipped"])
    mean = {v: {k: float(np.mean([r[v][k] for r in runs])) for k in names} for v in ("shipped", "composed")}
    worst = {v: float(np.mean([max(r[v].values()) for r in runs])) for v in mean}
    med = {v: float(np.mean([np.median(list(r[v].values())) for r in runs])) for v in mean}
    print("three-seed means: worst tensor shipped %.4f composed %.4f, median shipped %.4f composed %.4f" %
          (worst["shipped"], worst["composed"], med["shipped"], med["composed"]))
    ratios = sorted(((mean["shipped"][k] / max(mean["composed"][k], 1e-9), k) for k in names), reverse=True)
    for r_, k in ratios[:8]:
        print("    x%.2f  %.4f vs %.4f  %s" % (r_, mean["shipped"][k], mean["composed"][k], k))
    # measured on MI355X (round 4): three-seed means worst tensor 0.058 (shipped) / 0.053 (composed), median 0.0089 / 0.0052.
    # Round 4 found the tensors on the SCORE path of the decoder's encoder-decoder attention (q / k projections, the LayerNorm in
    # front) at 2.7 - 3.0 x the composed path's error: the fused backward took delta_i = sum_c dO_ic O_ic from the bf16-ROUNDED
    # attention output, the composed path sums P dP in fp32, and with 250 nearly uniform keys dS = P (dP - delta) cancels to a
    # fraction of either term.  Round 5: the forward hands the rounding remainder of O to the backward (o_lo, include/s2t_hip.h)
    # and delta is taken on O + o_lo — the bound is 2 x for EVERY tensor.  A wrong scale, a dropped bias gradient or a
    # mis-addressed row reads 1.0 on its tensor: a ratio of 50 - 100.
    for k in names:
        assert mean["shipped"][k] <= 2.0 * mean["composed"][k] + 0.01, (k, mean["shipped"][k], mean["composed"][k])
    # round 5 on this batch: three-seed means worst tensor 0.048 (shipped) / 0.049 (composed), median 0.0033 / 0.0033; the largest
    # per-tensor ratio x1.12
    assert worst["shipped"] < 0.11 and med["shipped"] < 0.012, (worst, med)
