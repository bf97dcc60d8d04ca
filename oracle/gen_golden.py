#!/usr/bin/env python3
"""Generate golden vectors from the reference itself (build container only).

TEST INFRASTRUCTURE.  Imports the reference (``/root/reference``, a fairseq fork) with the inert
third-party stand-ins of ``ref_stubs.py``, builds small reference models with seeded
weights and dumps inputs, weights and outputs as ``tests/golden/*.npz``.  The reference never travels
to the GPU box; these small ``.npz`` files (data only) do.

Run (from a cwd OUTSIDE the repo so that no by-product can land in either tree):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo/oracle \
        python /root/repo/oracle/gen_golden.py /root/repo/tests/golden

Reference entry points exercised (file:line under /root/reference/fairseq):
  models/speech_to_text/s2t_transformer.py:1714  S2TTransformerEncoder.forward
  models/transformer.py:1249                     TransformerDecoder.extract_features_scriptable
  criterions/label_smoothed_cross_entropy_with_ctc.py:74  joint CE + CTC loss
  models/speech_to_text/s2t_ctc.py:236           CTCDecoder.generate (greedy)
  modules/speech_to_text/subsampling.py:145      Conv1dSubsampling.forward
  modules/positional_encoding.py:151             RelPositionalEncoding.forward
  modules/sinusoidal_positional_embedding.py:60  SinusoidalPositionalEmbedding.forward
"""
import os
import sys
from argparse import Namespace

import numpy as np

import ref_stubs

ref_stubs.install()

import torch  # noqa: E402

import fairseq  # noqa: E402,F401
from fairseq.data import Dictionary  # noqa: E402
from fairseq.models import ARCH_CONFIG_REGISTRY, ARCH_MODEL_REGISTRY  # noqa: E402

torch.set_num_threads(8)
torch.backends.mkldnn.enabled = True


def make_dict(v):
    d = Dictionary()  # <s>=0 (CTC blank) <pad>=1 </s>=2 <unk>=3
    for i in range(v - 4):
        d.add_symbol("w%d" % i)
    assert len(d) == v
    return d


class FakeTask:
    def __init__(self, d):
        self.source_dictionary = d
        self.target_dictionary = d
        self.src_dict = d
        self.tgt_dict = d

    def get_source_dictionary(self, i):
        return self.source_dictionary


def base_args(arch, **kw):
    a = Namespace(
        arch=arch,
        input_feat_per_channel=80,
        input_channels=1,
        max_source_positions=6000,
        max_target_positions=1024,
        ctc_weight=0.3,
        layer_padding_mask=False,
        fp16=False,
        dropout=0.0,
        attention_dropout=0.0,
        activation_dropout=0.0,
        share_decoder_input_output_embed=True,
        share_ctc_and_embed=True,
        encoder_embed_norm=True,
        encoder_no_scale_embedding=True,
        subsampling_type="conv1d",
        subsampling_layers=2,
        subsampling_kernel=5,
        subsampling_stride=2,
        subsampling_norm="none",
        subsampling_activation="glu",
        activation_fn="relu",
    )
    for k, v in kw.items():
        setattr(a, k, v)
    ARCH_CONFIG_REGISTRY[arch](a)
    return a


def make_batch(B, T, V, seed, umin=3, umax=9):
    """collater-shaped batch: speech_to_text_dataset.py:411-485 (sorted desc, zero-padded,
    prev_output_tokens = eos moved to the front)."""
    g = torch.Generator().manual_seed(seed)
    lens = [T] + [int(torch.randint(int(0.6 * T), T + 1, (1,), generator=g)) for _ in range(B - 1)]
    lens = sorted(lens, reverse=True)
    src = torch.randn(B, T, 80, generator=g)
    for b, l in enumerate(lens):
        src[b, l:] = 0
    ulens = [int(torch.randint(umin, umax + 1, (1,), generator=g)) for _ in range(B)]
    U = max(ulens) + 1
    target = torch.full((B, U), 1, dtype=torch.long)
    prev = torch.full((B, U), 1, dtype=torch.long)
    for b, u in enumerate(ulens):
        toks = torch.randint(4, V, (u,), generator=g)
        target[b, :u] = toks
        target[b, u] = 2
        prev[b, 0] = 2
        prev[b, 1 : u + 1] = toks
    ntokens = int(sum(ulens) + B)
    return src, torch.tensor(lens), prev, target, ntokens


def seed_weights(model, seed):
    """Re-draw every parameter from a seeded N(0, s) so that no weight is left at a trivial init
    (LayerNorm gain 1 / bias 0 would hide a swapped gain/bias)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / fan_in**0.5))
            elif name.endswith("weight"):  # norm gains
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            if name.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))


def sd_np(model):
    out = {}
    for k, v in model.state_dict().items():
        out["w::" + k] = v.detach().cpu().numpy().copy()  # copy: BN buffers are updated in place later
    return out


def build(arch, V, **kw):
    d = make_dict(V)
    task = FakeTask(d)
    args = base_args(arch, **kw)
    model = ARCH_MODEL_REGISTRY[arch].build_model(args, task)
    return model, args, task


def np_(t):
    return t.detach().cpu().numpy()


def criterion_for(task, args):
    from fairseq.criterions.ctc import CtcCriterionConfig
    from fairseq.criterions.label_smoothed_cross_entropy_with_ctc import (
        LabelSmoothedCrossEntropyCriterionWithCTC,
    )

    cfg = CtcCriterionConfig()
    cfg.sentence_avg = False
    cfg.zero_infinity = True
    cfg.post_process = "none"
    cfg.inter_ctc_weight = float(getattr(args, "inter_ctc_weight", 0.0) or 0.0)
    crit = LabelSmoothedCrossEntropyCriterionWithCTC(
        task, label_smoothing=0.1, sentence_avg=False, cfg=cfg, ctc_weight=args.ctc_weight
    )
    return crit


def encdec_case(name, outdir, arch, V, B, T, seed, train_bn=False, tweak=None, **kw):
    torch.manual_seed(seed)
    model, args, task = build(arch, V, **kw)
    seed_weights(model, seed + 100)
    if tweak is not None:
        with torch.no_grad():
            tweak(model)
    src, lens, prev, target, ntokens = make_batch(B, T, V, seed + 200)
    out = {}
    out.update(sd_np(model))
    out["in::src_tokens"] = np_(src)
    out["in::src_lengths"] = np_(lens)
    out["in::prev_output_tokens"] = np_(prev)
    out["in::target"] = np_(target)
    out["in::ntokens"] = np.int64(ntokens)

    # ---- eval forward
    model.eval()
    with torch.no_grad():
        enc = model.encoder(src, lens)
        logits, extra = model.decoder(prev_output_tokens=prev, encoder_out=enc)
    out["out::encoder_out"] = np_(enc["encoder_out"][0])  # (T', B, d)
    out["out::ctc_logit"] = np_(enc["ctc_logit"][0])  # (T', B, V)
    out["out::encoder_padding_mask"] = np_(enc["encoder_padding_mask"][0])
    out["out::decoder_logits"] = np_(logits)  # (B, U, V)
    for i, il in enumerate(enc.get("inter_ctc_logits", [])):
        if isinstance(il, (list, tuple)) and il[1] is not None:
            out["out::inter_ctc_mask_%d" % i] = np_(il[1])

    # ---- loss + grads (training mode, dropout 0; BatchNorm uses batch statistics when present)
    model.train()
    crit = criterion_for(task, args)
    crit.train()
    sample = {
        "id": torch.arange(B),
        "net_input": {"src_tokens": src, "src_lengths": lens, "prev_output_tokens": prev},
        "target": target,
        "ntokens": ntokens,
    }
    model.zero_grad()
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    out["out::loss"] = np.float64(loss.item())
    out["out::trans_loss"] = np.float64(log["trans_loss"])
    out["out::nll_loss"] = np.float64(log["nll_loss"])
    out["out::ctc_loss"] = np.float64(log["ctc_loss"])
    if "inter_ctc_loss" in log:
        out["out::inter_ctc_loss"] = np.float64(log["inter_ctc_loss"])
        for i, il in enumerate(enc["inter_ctc_logits"]):
            out["out::inter_ctc_logit_%d" % i] = np_(il[0] if isinstance(il, (list, tuple)) else il)
    out["out::n_correct"] = np.int64(log["n_correct"])
    out["out::total"] = np.int64(log["total"])
    out["out::sample_size"] = np.int64(sample_size)
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["grad::" + k] = np_(p.grad)
    if train_bn:
        for k, b in model.named_buffers():
            if "running_" in k or "num_batches" in k:
                out["bn_after::" + k] = np_(b)
    # hyper-parameters the restatement needs: every scalar / string field of the resolved args
    for k, v in sorted(vars(args).items()):
        if isinstance(v, bool):
            out["cfg::" + k] = np.bool_(v)
        elif isinstance(v, (int, float)):
            out["cfg::" + k] = np.float64(v)
        elif isinstance(v, str):
            out["cfg::" + k] = np.array(v)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    print(name, "loss", loss.item(), {k: v for k, v in log.items() if "loss" in k})


EOS_LIFT = float(os.environ.get("EOS_LIFT", "9.0"))


def beam_case(name, outdir, V, B, T, seed, beam, max_len_b, sharpen, eos_lift, **kw):
    """fairseq/sequence_generator.py:191-614 (SequenceGenerator._generate, BeamSearch of search.py:101-150) on a small
    s2t_transformer with the incremental decoder (KV cache, modules/multihead_attention.py:302-339).  The output
    projection (untied from the input embedding, so that a token does not simply vote for itself) is sharpened so that
    candidate margins dwarf fp32 re-association noise."""
    from fairseq.sequence_generator import SequenceGenerator

    torch.manual_seed(seed)
    model, args, task = build("s2t_transformer_s", V, **kw)
    seed_weights(model, seed + 100)
    with torch.no_grad():
        model.decoder.output_projection.weight.mul_(sharpen)  # untied from the input embedding for this case
        # a constant lift of the </s> logit through the final LayerNorm's bias: hypotheses of different lengths
        bln = model.decoder.layer_norm.bias
        model.decoder.output_projection.weight[2].add_(eos_lift * bln / (bln * bln).sum())
    src, lens, _, _, _ = make_batch(B, T, V, seed + 200)
    model.eval()
    gen = SequenceGenerator([model], task.target_dictionary, beam_size=beam, max_len_a=0, max_len_b=max_len_b, min_len=1,
                            normalize_scores=True, len_penalty=1.0, unk_penalty=0.0)
    sample = {"id": torch.arange(B), "net_input": {"src_tokens": src, "src_lengths": lens}}
    with torch.no_grad():
        hyps = gen.generate([model], sample)
    out = {}
    out.update(sd_np(model))
    out["in::src_tokens"] = np_(src)
    out["in::src_lengths"] = np_(lens)
    out["gen::beam"] = np.int64(beam)
    out["gen::max_len_b"] = np.int64(max_len_b)
    out["gen::n_hyps"] = np.array([len(h) for h in hyps], dtype=np.int64)
    for b, hb in enumerate(hyps):
        for k, h in enumerate(hb):
            out["out::tokens_%d_%d" % (b, k)] = np_(h["tokens"])
            out["out::score_%d_%d" % (b, k)] = np.float64(float(h["score"]))
            out["out::pos_scores_%d_%d" % (b, k)] = np_(h["positional_scores"])
    for k, v in sorted(vars(args).items()):
        if isinstance(v, bool):
            out["cfg::" + k] = np.bool_(v)
        elif isinstance(v, (int, float)):
            out["cfg::" + k] = np.float64(v)
        elif isinstance(v, str):
            out["cfg::" + k] = np.array(v)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    print(name, [[h["tokens"].tolist() for h in hb[:2]] for hb in hyps], [float(hb[0]["score"]) for hb in hyps])


def specaug_case(outdir):
    """data/audio/feature_transforms/specaugment.py:79-131 with numpy's global RNG seeded: inputs, parameters, outputs."""
    from fairseq.data.audio.feature_transforms.specaugment import SpecAugmentTransform

    out = {}
    rng = np.random.RandomState(11)
    cases = [(0, 2, 27, 2, 40, 0.5, 0.0), (1, 1, 13, 3, 100, 0.2, None), (2, 2, 27, 2, 100, 1.0, 0.0)]
    for i, (seed, fn, ff, tn, tt, tp, mv) in enumerate(cases):
        x = rng.randn(60 + 70 * i, 80).astype(np.float32) * 2 + 0.5
        tr = SpecAugmentTransform(0, fn, ff, tn, tt, tp, mv)
        np.random.seed(100 + seed)
        y = tr(x)
        out["in::x_%d" % i] = x
        out["out::y_%d" % i] = y
        out["cfg::params_%d" % i] = np.array([100 + seed, fn, ff, tn, tt, tp, -1.0 if mv is None else mv], dtype=np.float64)
    out["cfg::n"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(outdir, "specaugment.npz"), **out)
    print("specaugment", [int((out["out::y_%d" % i] != out["in::x_%d" % i]).sum()) for i in range(len(cases))])


def ctc_greedy_case(name, outdir, V, B, T, seed, **kw):
    from fairseq.models.speech_to_text.s2t_ctc import CTCDecoder

    torch.manual_seed(seed)
    model, args, task = build("s2t_ctc_s", V, **kw)
    seed_weights(model, seed + 100)
    # make the greedy output non-trivial: scale the CTC projection so that the arg-max changes over time
    with torch.no_grad():
        model.encoder.ctc.ctc_projection.weight.mul_(4.0)
    src, lens, prev, target, ntokens = make_batch(B, T, V, seed + 200)
    model.eval()
    dargs = Namespace(beam=1, ctc_self_ensemble=False, ctc_inter_logit=0, cal_flops=False, print_alignment=False)
    dec = CTCDecoder([model], dargs, task.target_dictionary, blank_idx=0)
    sample = {"net_input": {"src_tokens": src, "src_lengths": lens, "prev_output_tokens": prev}, "target": target}
    with torch.no_grad():
        hyps = dec.generate([model], sample)
        enc = model.encoder(src, lens)
    out = {}
    out.update(sd_np(model))
    out["in::src_tokens"] = np_(src)
    out["in::src_lengths"] = np_(lens)
    out["out::ctc_logit"] = np_(enc["ctc_logit"][0])
    out["out::encoder_padding_mask"] = np_(enc["encoder_padding_mask"][0])
    toks = [np_(h[0]["tokens"]).astype(np.int64) for h in hyps]
    out["out::hyp_lengths"] = np.array([len(t) for t in toks], dtype=np.int64)
    out["out::hyp_tokens"] = np.concatenate(toks) if sum(len(t) for t in toks) else np.zeros(0, np.int64)
    out["out::hyp_scores"] = np.array([float(h[0]["score"]) for h in hyps])
    for k in "encoder_embed_dim encoder_ffn_embed_dim encoder_layers encoder_attention_heads subsampling_filter".split():
        out["cfg::" + k] = np.float64(getattr(args, k))
    for k in "encoder_attention_type encoder_activation_fn activation_fn".split():
        out["cfg::" + k] = np.array(getattr(args, k))
    for k in "macaron_style use_cnn_module layer_padding_mask encoder_normalize_before".split():
        out["cfg::" + k] = np.bool_(bool(getattr(args, k)))
    out["cfg::cnn_module_kernel"] = np.float64(getattr(args, "cnn_module_kernel", 0))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    print(name, "hyp lens", [len(t) for t in toks])


def _cfg_dump(out, args):
    for k, v in sorted(vars(args).items()):
        if isinstance(v, bool):
            out["cfg::" + k] = np.bool_(v)
        elif isinstance(v, (int, float)):
            out["cfg::" + k] = np.float64(v)
        elif isinstance(v, str):
            out["cfg::" + k] = np.array(v)


def nast_case(name, outdir, V, B, T, seed, **kw):
    """SURVEY.md §8f row 2 — the reproduction_nast.yaml stack at toy size: ``s2t_ctc --encoder-type sate`` (models/
    speech_to_text/s2t_ctc.py:28-126 -> S2TSATEEncoder s2t_sate.py:837-1075, TextualEncoder :333-827 with
    TransformerS2EncoderLayer modules/transformer_s2_layer.py:214-336, XCTC / inter-XCTC heads, PAE adapter.py:189-297),
    CtcCriterion (criterions/ctc.py:258-1016) and CTCDecoder greedy on xctc_logit (s2t_ctc.py:236-349).
    The reference's random draws in the training pass (``torch.rand`` of the PAE ground-truth mask, numpy ``uniform`` of the
    league drop-net) are recorded so that the restatement can replay them."""
    from fairseq.criterions.ctc import CtcCriterion, CtcCriterionConfig
    from fairseq.models.speech_to_text.s2t_ctc import CTCDecoder
    import fairseq.modules.transformer_s2_layer as s2mod

    torch.manual_seed(seed)
    model, args, task = build("s2t_ctc", V, **kw)
    seed_weights(model, seed + 100)
    with torch.no_grad():  # non-trivial greedy output / alignments
        model.encoder.textual_encoder.xctc.ctc_projection.weight.mul_(3.0)
        model.encoder.acoustic_encoder.ctc.ctc_projection.weight.mul_(3.0)
    src, lens, prev, target, ntokens = make_batch(B, T, V, seed + 200, umin=3, umax=7)
    _, _, _, transcript, _ = make_batch(B, T, V, seed + 300, umin=3, umax=7)
    out = {}
    out.update(sd_np(model))
    out["in::src_tokens"], out["in::src_lengths"] = np_(src), np_(lens)
    out["in::prev_output_tokens"], out["in::target"], out["in::transcript"] = np_(prev), np_(target), np_(transcript)
    out["in::ntokens"] = np.int64(ntokens)

    model.eval()
    with torch.no_grad():
        enc = model.encoder(src, lens)
        dargs = Namespace(beam=1, ctc_self_ensemble=False, ctc_inter_logit=0, cal_flops=False, print_alignment=False)
        hyps = CTCDecoder([model], dargs, task.target_dictionary, blank_idx=0).generate(
            [model], {"net_input": {"src_tokens": src, "src_lengths": lens, "prev_output_tokens": prev}, "target": target})
    out["out::encoder_out"] = np_(enc["encoder_out"][0])
    out["out::ctc_logit"] = np_(enc["ctc_logit"][0])
    out["out::xctc_logit"] = np_(enc["xctc_logit"][0])
    out["out::encoder_padding_mask"] = np_(enc["encoder_padding_mask"][0])
    for i, il in enumerate(enc["inter_ctc_logits"]):
        out["out::inter_ctc_logit_%d" % i] = np_(il[0] if isinstance(il, (list, tuple)) else il)
    for i, il in enumerate(enc["inter_xctc_logits"]):
        out["out::inter_xctc_logit_%d" % i] = np_(il[0] if isinstance(il, (list, tuple)) else il)
    toks = [np_(h[0]["tokens"]).astype(np.int64) for h in hyps]
    out["out::hyp_lengths"] = np.array([len(t) for t in toks], dtype=np.int64)
    out["out::hyp_tokens"] = np.concatenate(toks) if sum(len(t) for t in toks) else np.zeros(0, np.int64)
    out["out::hyp_scores"] = np.array([float(h[0]["score"]) for h in hyps])

    # ---- CtcCriterion in training mode (dropout 0), random draws recorded
    cfg = CtcCriterionConfig()
    cfg.sentence_avg, cfg.zero_infinity, cfg.post_process = False, True, "none"
    cfg.xctc_weight = float(args.xctc_weight)
    cfg.inter_ctc_weight = float(getattr(args, "inter_ctc_weight", 0.0))
    cfg.inter_xctc_weight = float(getattr(args, "inter_xctc_weight", 0.0))
    crit = CtcCriterion(cfg, task, ctc_weight=float(args.ctc_weight))
    model.train()
    crit.train()
    sample = {"id": torch.arange(B), "net_input": {"src_tokens": src, "src_lengths": lens, "prev_output_tokens": prev},
              "target": target, "transcript": {"tokens": transcript}, "ntokens": ntokens}
    rand_draws, unif_draws = [], []
    real_rand, real_unif = torch.rand, s2mod.uniform

    def rec_rand(*a, **k):
        r = real_rand(*a, **k)
        rand_draws.append(r.clone())
        return r

    def rec_unif(*a, **k):
        r = real_unif(*a, **k)
        unif_draws.append(float(r))
        return r

    # criterions/ctc.py:290 calls torch_imputer's best_alignment, a CUDA extension that cannot be built or run in this
    # container (fairseq.torch_imputer is blocked by ref_stubs).  For the curriculum fixture the Viterbi alignment is
    # therefore supplied by the oracle's restatement (oracle/s2t_oracle.py: best_alignment); everything downstream of it
    # (oracle labels, mistake flags, masks, PAE mixing, losses, gradients) is the reference's own code.
    import fairseq.criterions.ctc as ctc_mod
    import s2t_oracle as ORC

    def best_alignment_standin(log_prob, targets, input_lengths, target_lengths, blank=0, zero_infinity=False):
        return ORC.best_alignment(log_prob, [targets[b][: int(target_lengths[b])].tolist() for b in range(targets.size(0))],
                                  input_lengths, blank)

    had = getattr(ctc_mod, "best_alignment", None)
    ctc_mod.best_alignment = best_alignment_standin
    torch.rand, s2mod.uniform = rec_rand, rec_unif
    # the league drop-net draws come from numpy's GLOBAL generator (modules/transformer_s2_layer.py: numpy.random.uniform):
    # seed it, and torch's, here so that the recorded draws — and with them the whole fixture — regenerate bit for bit
    np.random.seed(seed + 400)
    torch.manual_seed(seed + 500)
    try:
        model.zero_grad()
        loss, sample_size, log = crit(model, sample)
        loss.backward()
    finally:
        torch.rand, s2mod.uniform = real_rand, real_unif
        if had is None:
            del ctc_mod.best_alignment
        else:
            ctc_mod.best_alignment = had
    out["out::loss"] = np.float64(loss.item())
    for k in ("ctc_loss", "inter_ctc_loss", "xctc_loss", "inter_xctc_loss"):
        if k in log:
            out["out::" + k] = np.float64(log[k])
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["grad::" + k] = np_(p.grad)
    for k, b in model.named_buffers():
        if "running_" in k:
            out["bn_after::" + k] = np_(b)
    gt = float(getattr(args, "xctc_pae_ground_truth_ratio", 0) or 0)
    if gt > 0:
        big = [r for r in rand_draws if r.numel() > 1]
        assert len(big) == 1, [tuple(r.shape) for r in rand_draws]
        out["aux::xctc_rand_mask"] = np_(big[0] < gt)
    # the criterion runs the encoder twice when gt > 0 (alignment pass first): keep the draws of the LAST pass
    n_s2 = int(args.text_encoder_layers) - int(args.cross_attn_start_layer) + 1
    prob = float(args.cross_attn_league_drop_net_prob)
    out["aux::drop_self_attn_all"] = np.array([u < prob for u in unif_draws], dtype=np.bool_)
    out["aux::n_s2_layers"] = np.int64(n_s2)
    _cfg_dump(out, args)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    print(name, "loss", loss.item(), {k: v for k, v in log.items() if "loss" in k}, "hyps", [len(t) for t in toks],
          "uniform draws", len(unif_draws), "rand draws", [tuple(r.shape) for r in rand_draws])


def dataset_case(outdir):
    """SURVEY.md §8 rows a2/a3 and f3 (on-disk format): a small dataset in the reference's own format — TSV manifest
    (speech_to_text_dataset.py:594-652), features as .npy members of an UNCOMPRESSED zip addressed by byte offset
    (:193-264; written like examples/speech_to_text/data_utils.py:101-126), config yaml, dictionary — loaded with the
    reference's SpeechToTextDatasetCreator.from_tsv; items, ``ordered_indices`` and collated batches (:411-485) are dumped.
    The dataset directory itself (tests/golden/s2t_dataset/) is data written by this script."""
    import zipfile
    from fairseq.data.audio.speech_to_text_dataset import S2TDataConfig, SpeechToTextDatasetCreator

    root = os.path.join(outdir, "s2t_dataset")
    os.makedirs(root, exist_ok=True)
    rng = np.random.RandomState(3)
    n_utt, V = 7, 24
    words = ["w%d" % i for i in range(V)]
    with open(os.path.join(root, "dict.txt"), "w") as f:
        for i, w in enumerate(words):
            f.write("%s %d\n" % (w, 100 - i))
    feats, texts = [], []
    for i in range(n_utt):
        n = int(rng.randint(20, 60))
        feats.append((rng.randn(n, 80) * (1 + 0.3 * i) + 0.2 * i).astype(np.float32))
        toks = [words[int(t)] for t in rng.randint(0, V, size=int(rng.randint(2, 7)))]
        if i == 2:
            toks[1] = "oov"  # -> <unk>
        texts.append(" ".join(toks))
    zpath = os.path.join(root, "fbank80.zip")
    with zipfile.ZipFile(zpath, "w", zipfile.ZIP_STORED) as zf:
        for i, x in enumerate(feats):
            tmp = os.path.join(root, "utt%d.npy" % i)
            np.save(tmp, x)
            zf.write(tmp, arcname="utt%d.npy" % i)
            os.remove(tmp)
    with zipfile.ZipFile(zpath, "r") as zf:
        info = {i.filename: (i.header_offset + 30 + len(i.filename), i.file_size) for i in zf.infolist()}
    np.save(os.path.join(root, "utt_plain.npy"), feats[0])  # one plain .npy path as well
    with open(os.path.join(root, "train.tsv"), "w") as f:
        f.write("id\taudio\tn_frames\ttgt_text\tspeaker\n")
        for i in range(n_utt):
            off, size = info["utt%d.npy" % i]
            audio = "utt_plain.npy" if i == 0 else "fbank80.zip:%d:%d" % (off, size)
            f.write("utt%d\t%s\t%d\t%s\tspk%d\n" % (i, audio, feats[i].shape[0], texts[i], i % 2))
    allf = np.concatenate(feats, 0)
    np.savez(os.path.join(root, "gcmvn.npz"), mean=allf.mean(0), std=allf.std(0))
    for name, extra in (("config_utt.yaml", "transforms:\n  '*': [utterance_cmvn]\n"),
                        ("config_global.yaml", "transforms:\n  '*': [global_cmvn]\ncmvn: global\ncmvn_path: gcmvn.npz\n")):
        with open(os.path.join(root, name), "w") as f:
            f.write("audio_root: %s\ninput_feat_per_channel: 80\ninput_channels: 1\nvocab_filename: dict.txt\n%s" % ("AUDIO_ROOT", extra))
    out = {}
    d = Dictionary.load(os.path.join(root, "dict.txt"))
    for cfg_name in ("config_utt.yaml", "config_global.yaml"):
        # the manifest's audio paths are relative: point audio_root at this checkout for the reference run
        tmp_cfg = os.path.join(root, "_tmp_" + cfg_name)
        with open(os.path.join(root, cfg_name)) as f, open(tmp_cfg, "w") as g:
            g.write(f.read().replace("AUDIO_ROOT", root))
        cfg = S2TDataConfig(tmp_cfg)
        ds = SpeechToTextDatasetCreator.from_tsv(root, cfg, "train", d, None, None, is_train_split=False, epoch=1, seed=1)
        os.remove(tmp_cfg)
        tag = cfg_name.split(".")[0]
        out[tag + "::ordered_indices"] = np.asarray(ds.ordered_indices(), dtype=np.int64)
        idx = [4, 0, 6, 2, 5]
        items = [ds[i] for i in idx]
        for i, it in zip(idx, items):
            out[tag + "::item_%d_source" % i] = np_(it[1])
            out[tag + "::item_%d_target" % i] = np_(it[2])
        b = ds.collater(items)
        out[tag + "::batch_idx"] = np.array(idx, dtype=np.int64)
        out[tag + "::id"] = np_(b["id"])
        out[tag + "::src_tokens"] = np_(b["net_input"]["src_tokens"])
        out[tag + "::src_lengths"] = np_(b["net_input"]["src_lengths"])
        out[tag + "::prev_output_tokens"] = np_(b["net_input"]["prev_output_tokens"])
        out[tag + "::target"] = np_(b["target"])
        out[tag + "::target_lengths"] = np_(b["target_lengths"])
        out[tag + "::ntokens"] = np.int64(b["ntokens"])
        out[tag + "::nsentences"] = np.int64(b["nsentences"])
        out[tag + "::sizes"] = np.asarray(ds.sizes, dtype=np.int64)
    np.savez_compressed(os.path.join(outdir, "s2t_dataset_expected.npz"), **out)
    print("dataset ok", out["config_utt::src_tokens"].shape, out["config_utt::target"].tolist())


def module_cases(outdir):
    from fairseq.modules import LayerNorm
    from fairseq.modules.positional_encoding import RelPositionalEncoding
    from fairseq.modules.sinusoidal_positional_embedding import SinusoidalPositionalEmbedding
    from fairseq.modules.speech_to_text.subsampling import Conv1dSubsampling
    from fairseq.data.data_utils import lengths_to_padding_mask

    out = {}
    g = torch.Generator().manual_seed(7)
    # Conv1dSubsampling(2, 80, [48, 32], 5, 2, "none", "glu"); its nn.Conv1d initialisers draw from the GLOBAL generator
    torch.manual_seed(70)
    sub = Conv1dSubsampling(2, 80, [48, 32], 5, stride=2, norm="none", act="glu")
    x = torch.randn(37, 3, 80, generator=g)
    lens = torch.tensor([37, 30, 22])
    with torch.no_grad():
        y, yl = sub(x, lens)
    for k, v in sub.state_dict().items():
        out["sub::w::" + k] = np_(v)
    out["sub::x"] = np_(x)
    out["sub::lens"] = np_(lens)
    out["sub::y"] = np_(y)
    out["sub::ylens"] = np_(yl)
    # rel-pos table
    rp = RelPositionalEncoding(6000, 32)
    out["relpos::T7"] = np_(rp(torch.zeros(7, 1, 32)))
    out["relpos::T12"] = np_(rp(torch.zeros(12, 1, 32)))
    # sinusoidal table driven by the bool padding mask (s2t_transformer.py:1785)
    sp = SinusoidalPositionalEmbedding(32, 1, init_size=64)
    mask = lengths_to_padding_mask(torch.tensor([9, 6, 4]))
    out["sinpos::mask"] = np_(mask)
    out["sinpos::out"] = np_(sp(mask))
    toks = torch.tensor([[2, 5, 6, 7, 1], [2, 9, 1, 1, 1]])
    out["sinpos::tokens"] = np_(toks)
    out["sinpos::tokens_out"] = np_(sp(toks))
    # LayerNorm
    ln = LayerNorm(32)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(32, generator=g))
        ln.bias.copy_(0.1 * torch.randn(32, generator=g))
    xx = torch.randn(5, 32, generator=g)
    out["ln::x"] = np_(xx)
    out["ln::w"] = np_(ln.weight)
    out["ln::b"] = np_(ln.bias)
    out["ln::y"] = np_(ln(xx))
    np.savez_compressed(os.path.join(outdir, "modules.npz"), **out)
    print("modules ok")


def trainer_case(name, outdir, arch, V, B, T, seed, updates=5, **kw):
    """Row a22 pinned to the reference: ``updates`` optimizer updates of a small model with the reference's own classes in
    the order of Trainer.train_step (fairseq/trainer.py:611-759): zero_grad -> criterion forward/backward -> [all-reduce:
    one rank, LegacyDDP's pre-division by 1] -> multiply_grads(world / sample_size) (:729-734) -> clip_grad_norm
    (fairseq/utils.py:328-369 through optim/fairseq_optimizer.py:109-111) -> FairseqAdam.step (optim/adam.py:146-226) ->
    lr_scheduler.step_update(num_updates) (optim/lr_scheduler/inverse_square_root_schedule.py:79-85, from trainer.py:802).
    Recipe hyper-parameters (egs/mustc/asr/conf/base.yaml:4-9) with a short warm-up so that the schedule moves."""
    from fairseq.optim.adam import FairseqAdam
    from fairseq.optim.lr_scheduler.inverse_square_root_schedule import InverseSquareRootSchedule

    torch.manual_seed(seed)
    model, args, task = build(arch, V, **kw)
    seed_weights(model, seed + 100)
    src, lens, prev, target, ntokens = make_batch(B, T, V, seed + 200)
    out = {}
    out.update(sd_np(model))
    out["in::src_tokens"], out["in::src_lengths"] = np_(src), np_(lens)
    out["in::prev_output_tokens"], out["in::target"], out["in::ntokens"] = np_(prev), np_(target), np.int64(ntokens)
    hp = dict(lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, clip_norm=10.0, warmup_updates=3, warmup_init_lr=1e-7)
    ocfg = Namespace(lr=[hp["lr"]], adam_betas=str(hp["betas"]), adam_eps=hp["eps"], weight_decay=hp["weight_decay"],
                     use_old_adam=True, tpu=False)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FairseqAdam(ocfg, params)
    scfg = Namespace(lr=[hp["lr"]], warmup_updates=hp["warmup_updates"], warmup_init_lr=hp["warmup_init_lr"])
    sched = InverseSquareRootSchedule(scfg, opt)
    sched.step_update(0)  # trainer.py:_build_optimizer -> lr_step_update(0)
    crit = criterion_for(task, args)
    model.train()
    crit.train()
    sample = {"id": torch.arange(B), "net_input": {"src_tokens": src, "src_lengths": lens, "prev_output_tokens": prev},
              "target": target, "ntokens": ntokens}
    losses, gnorms, lrs = [], [], []
    for n in range(updates):
        opt.zero_grad()
        loss, sample_size, log = crit(model, sample)
        opt.backward(loss)
        opt.multiply_grads(1.0 / float(sample_size))  # one rank: world / sample_size
        gnorm = opt.clip_grad_norm(hp["clip_norm"])
        lrs.append(opt.get_lr())  # the rate this update runs with
        opt.step()
        sched.step_update(n + 1)
        losses.append(float(loss.item()))
        gnorms.append(float(gnorm))
    out["out::loss"], out["out::gnorm"], out["out::lr"] = np.array(losses), np.array(gnorms), np.array(lrs)
    for k, v in model.state_dict().items():
        out["after::" + k] = v.detach().cpu().numpy().copy()
    for k, v in hp.items():
        out["hp::" + k] = np.array(v, dtype=np.float64)
    for k, v in sorted(vars(args).items()):
        if isinstance(v, bool):
            out["cfg::" + k] = np.bool_(v)
        elif isinstance(v, (int, float)):
            out["cfg::" + k] = np.float64(v)
        elif isinstance(v, str):
            out["cfg::" + k] = np.array(v)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    print(name, "losses", losses, "gnorm", gnorms, "lr", lrs)


def check(committed):
    """``gen_golden.py --check <dir>``: regenerate every fixture (or the GOLDEN_ONLY group) into a temporary directory and
    compare with the committed ``.npz`` files key by key.  fairseq's uninitialised ``_float_tensor`` buffers are skipped.
    Exit status 1 on any difference beyond 1e-6 of a tensor's scale (threading can move the last bits of a CPU GEMM)."""
    import glob
    import tempfile

    tmp = tempfile.mkdtemp(prefix="golden_check_")
    sys.argv[1] = tmp
    main()
    bad = 0
    for f in sorted(glob.glob(os.path.join(tmp, "*.npz"))):
        name = os.path.basename(f)
        ref = os.path.join(committed, name)
        if not os.path.exists(ref):
            print("MISSING in", committed, ":", name)
            bad += 1
            continue
        a, b = np.load(f, allow_pickle=True), np.load(ref, allow_pickle=True)
        keys_a = {k for k in a.files if "_float_tensor" not in k}
        keys_b = {k for k in b.files if "_float_tensor" not in k}
        if keys_a != keys_b:
            print("KEYS differ in", name, sorted(keys_a ^ keys_b)[:8])
            bad += 1
        worst, worst_key = 0.0, None
        for k in sorted(keys_a & keys_b):
            x, y = a[k], b[k]
            if x.shape != y.shape or x.dtype != y.dtype:
                print("SHAPE/DTYPE differs:", name, k, x.shape, y.shape, x.dtype, y.dtype)
                bad += 1
                continue
            if x.dtype.kind in "fc":
                scale = max(float(np.abs(y).max()) if y.size else 0.0, 1e-30)
                d = float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()) / scale if y.size else 0.0
            else:
                d = 0.0 if np.array_equal(x, y) else 1.0
            if d > worst:
                worst, worst_key = d, k
        print("%-34s %d keys, worst relative difference %.2e%s" % (name, len(keys_a & keys_b), worst, (" (" + worst_key + ")") if worst_key else ""))
        if worst > 1e-6:
            bad += 1
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    print("GOLDEN_CHECK", "FAILED" if bad else "OK")
    sys.exit(1 if bad else 0)


def main():
    outdir = sys.argv[1]
    os.makedirs(outdir, exist_ok=True)
    small = dict(
        encoder_embed_dim=32,
        encoder_ffn_embed_dim=64,
        encoder_attention_heads=2,
        decoder_attention_heads=2,
        encoder_layers=2,
        decoder_layers=2,
        subsampling_filter=48,
    )
    conf = dict(
        macaron_style=True,
        use_cnn_module=True,
        cnn_module_kernel=15,
        encoder_attention_type="rel_pos",
        encoder_activation_fn="swish",
        layer_padding_mask=True,
    )
    nast = dict(small, encoder_layers=4, encoder_type="sate", text_encoder_layers=4, acoustic_encoder="transformer",
                adapter="inter_league", xctc_weight=1.0, ctc_weight=1.0, share_ctc_and_embed=True, share_xctc_and_embed=True,
                text_no_pos_emb=True, textual_encoder_embed_norm=False, textual_encoder_no_scale_embedding=True,
                encoder_normalize_before=True, share_inter_ctc=True, inter_ctc_weight=1.0, inter_ctc_layers="2,3",
                inter_xctc_weight=1.0, inter_xctc_layers="2,3", ctc_pae="inter_league", xctc_pae="inter_league",
                xctc_cross_attn=True, cross_attn_start_layer=3, cross_attn_layer=2, cross_attn_collaboration_mode="serial",
                cross_attn_league_drop_net=True, xctc_pae_ground_truth_only_mistake=True, pae_oracle_smooth=True)
    if os.environ.get("GOLDEN_ONLY", "") in ("", "trainer"):
        trainer_case("trainer_conformer_small", outdir, "s2t_transformer_s", V=40, B=3, T=50, seed=2, **small, **conf)
    if os.environ.get("GOLDEN_ONLY", "") == "trainer":
        return
    if os.environ.get("GOLDEN_ONLY", "") in ("", "dataset"):
        dataset_case(outdir)
    if os.environ.get("GOLDEN_ONLY", "") == "dataset":
        return
    if os.environ.get("GOLDEN_ONLY", "") in ("", "compress"):
        # egs/*/conf/dynamic.yaml on top of inter.yaml: CTC-guided compression after the intermediate CTC layers.
        # The blank logit is lifted so that the blank posterior straddles the threshold (seeded weights alone would
        # keep every frame); ragged batch, Conformer (relative positions re-derived) and Transformer (absolute positions
        # re-added) variants.
        def lift_blank(m, lift=3.5):
            m.encoder.ctc.ctc_projection.weight.mul_(4.0)
            m.encoder.ctc.ctc_projection.bias[0] += lift

        comp = dict(inter_ctc_layers="2,3", share_inter_ctc=True, inter_ctc_weight=0.2, ctc_pae="none",
                    compression_metric="threshold", compression_mode="create", compression_layers="2,3",
                    compression_threshold="0.5", compression_norm=True, compression_pos=True)
        encdec_case("conformer_compress", outdir, "s2t_transformer_s", V=40, B=4, T=90, seed=31, train_bn=True,
                    tweak=lift_blank, **dict(small, encoder_layers=4), **conf, **comp)
        encdec_case("transformer_compress", outdir, "s2t_transformer_s", V=40, B=4, T=90, seed=32,
                    tweak=lambda m: lift_blank(m, float(os.environ.get("BLANK_LIFT", "7.0"))),
                    **dict(small, encoder_layers=4), **dict(comp, ctc_pae="inter_league"))
        # bf16-comparable variants: the SAME recipe with seeds and a sharpened blank logit chosen so that no frame's blank
        # log-odds at a compression layer comes closer than MARGIN to the threshold's (eval and training forward alike):
        # bf16 rounding then moves no frame across the threshold and the compressed tensors stay comparable frame by frame.
        # (bf16 rounding is RELATIVE, so sharpening the posterior does not help: a frame is at risk when its blank log-odds
        # lies within a few percent of the spread of the log-odds from the threshold's.  With ~1 % of the frames at risk
        # per draw, a small batch and a search over seeds finds draws with none.)
        MB, MT = 2, 48

        def margin_of(seed, tweak, kw):
            torch.manual_seed(seed)
            model, args, task = build("s2t_transformer_s", 40, **kw)
            seed_weights(model, seed + 100)
            with torch.no_grad():
                tweak(model)
            src, lens, prev, target, ntokens = make_batch(MB, MT, 40, seed + 200)
            worst, kept = 1e9, []
            for mode in ("eval", "train"):
                model.train(mode == "train")
                with torch.no_grad():
                    enc = model.encoder(src, lens)
                for il in enc["inter_ctc_logits"]:
                    logit, mask = il[0].float(), il[1]  # (T, B, V), (B, T)
                    odds = logit[:, :, 0] - torch.logsumexp(logit[:, :, 1:], dim=-1)  # log p/(1-p); threshold 0.5 <-> 0
                    valid = ~mask.transpose(0, 1)
                    o = odds[valid]
                    worst = min(worst, float(o.abs().min()))
                    kept.append(float((o < 0).float().mean()))
            return worst, kept

        MARGIN = 0.45  # |log-odds| >= 0.45: the blank posterior of every frame is outside [0.39, 0.61], >= 0.11 from the threshold
        for name, base_seed, kwm, lift, train_bn in (
                ("conformer_compress_margin", 131, dict(**dict(small, encoder_layers=4), **conf, **comp), 3.5, True),
                ("transformer_compress_margin", 231, dict(**dict(small, encoder_layers=4), **dict(comp, ctc_pae="inter_league")),
                 7.0, False)):
            def sharpen(m, lift=lift):
                m.encoder.ctc.ctc_projection.weight.mul_(4.0)
                m.encoder.ctc.ctc_projection.bias[0] += lift
            found = None
            best = 0.0
            for seed in range(base_seed, base_seed + 1500):
                worst, kept = margin_of(seed, sharpen, kwm)
                ok = all(0.45 < k < 0.96 for k in kept)
                if ok and worst > best:
                    best = worst
                    print(name, "seed", seed, "min |log-odds| %.3f" % worst, "kept", ["%.2f" % k for k in kept], flush=True)
                if worst >= MARGIN and ok:
                    found = seed
                    break
            assert found is not None, name
            encdec_case(name, outdir, "s2t_transformer_s", V=40, B=MB, T=MT, seed=found, train_bn=train_bn, tweak=sharpen, **kwm)
    if os.environ.get("GOLDEN_ONLY", "") == "compress":
        return
    if os.environ.get("GOLDEN_ONLY", "") in ("", "nast"):
        # egs/mustc/st/conf/reproduction_nast.yaml at toy size; first without, then with the curriculum randomness
        nast_case("nast_small", outdir, V=40, B=3, T=60, seed=21, **nast, **conf, cross_attn_league_drop_net_prob=0.0,
                  xctc_pae_ground_truth_ratio=0.0)
        nast_case("nast_pae_oracle", outdir, V=40, B=3, T=60, seed=22, **nast, **conf, cross_attn_league_drop_net_prob=0.5,
                  xctc_pae_ground_truth_ratio=0.8)
    if os.environ.get("GOLDEN_ONLY", "") == "nast":
        return
    if os.environ.get("GOLDEN_ONLY", "") in ("", "interctc"):
        # egs/mustc/asr/conf/inter.yaml: intermediate CTC heads sharing the top projection, own LayerNorms
        encdec_case("conformer_interctc", outdir, "s2t_transformer_s", V=40, B=3, T=50, seed=12, train_bn=True,
                    **dict(small, encoder_layers=4), **conf, inter_ctc_layers="2,3", share_inter_ctc=True,
                    inter_ctc_weight=0.2, ctc_pae="none")
    if os.environ.get("GOLDEN_ONLY", "") == "interctc":
        return
    if os.environ.get("GOLDEN_ONLY", "") in ("", "specaug"):
        specaug_case(outdir)
    if os.environ.get("GOLDEN_ONLY", "") == "specaug":
        return
    if os.environ.get("GOLDEN_ONLY", "") in ("", "beam"):
        beam_case("beam_search_transformer", outdir, V=40, B=3, T=50, seed=9, beam=4, max_len_b=12, sharpen=6.0, eos_lift=EOS_LIFT,
                  share_decoder_input_output_embed=False, share_ctc_and_embed=False, **small)
    if os.environ.get("GOLDEN_ONLY", "") == "beam":
        return
    if os.environ.get("GOLDEN_ONLY", "") in ("", "modules"):
        module_cases(outdir)
    if os.environ.get("GOLDEN_ONLY", "") == "modules":
        return
    encdec_case("transformer_small", outdir, "s2t_transformer_s", V=40, B=3, T=50, seed=1, **small)
    encdec_case("conformer_small", outdir, "s2t_transformer_s", V=40, B=3, T=50, seed=2, train_bn=True, **small, **conf)
    # ragged: one full row + short rows; T not a multiple of 4; B=4
    encdec_case("conformer_ragged", outdir, "s2t_transformer_s", V=37, B=4, T=67, seed=3, train_bn=True, **small, **conf)
    pds = dict(
        encoder_embed_dim=32, encoder_ffn_embed_dim=64, encoder_attention_heads=2, decoder_attention_heads=2,
        decoder_embed_dim=32, decoder_ffn_embed_dim=64, encoder_layers=4, decoder_layers=2,
        pds_stages=4, pds_layers="1_1_1_1", pds_ratios="2_2_1_2", pds_fusion=False, pds_embed_dims="32_32_32_32",
        pds_ds_method="conv", pds_embed_norm=True, pds_position_embed="1_1_1_1", pds_kernel_sizes="5_5_5_5",
        pds_ffn_ratios="2_2_2_2", pds_attn_heads="2_2_2_2",
    )
    for k in ("subsampling_type", "subsampling_layers", "subsampling_kernel", "subsampling_stride", "subsampling_norm",
              "subsampling_activation", "encoder_embed_norm", "encoder_no_scale_embedding"):
        pass
    if os.environ.get("GOLDEN_ONLY", "") in ("", "pds"):
        encdec_case("pds_small", outdir, "pdss2t_transformer_s_8", V=40, B=3, T=67, seed=6, **pds)
        encdec_case("pds_conformer_small", outdir, "pdss2t_transformer_s_8", V=40, B=3, T=64, seed=7, train_bn=True, **pds, **conf)
    if os.environ.get("GOLDEN_ONLY", "") in ("", "pdsfusion"):
        # pds_base_8.yaml's fusion settings switched on (the recipes ship them with pds-fusion False)
        encdec_case("pds_fusion_small", outdir, "pdss2t_transformer_s_8", V=40, B=3, T=67, seed=41, train_bn=True,
                    **dict(pds, pds_fusion=True), pds_fusion_method="all_conv2", pds_fusion_layers="0_1_1_1",
                    pds_fusion_weight="0.2_0.3_0.5")
        # the same with LEARNED fusion weights (no --pds-fusion-weight: nn.Parameter fusion_weight, :799-806)
        encdec_case("pds_fusion_learned", outdir, "pdss2t_transformer_s_8", V=40, B=3, T=67, seed=43, train_bn=True,
                    **dict(pds, pds_fusion=True), pds_fusion_method="all_conv2", pds_fusion_layers="0_1_1_1")
    if os.environ.get("GOLDEN_ONLY", "") == "pdsfusion":
        return
    sate = dict(small, text_encoder_layers=2, acoustic_encoder="transformer", adapter="inter_league",
                textual_encoder_embed_norm=True, textual_encoder_no_scale_embedding=True, encoder_normalize_before=True,
                decoder_normalize_before=True)
    if os.environ.get("GOLDEN_ONLY", "") in ("", "sate"):
        encdec_case("sate_small", outdir, "s2t_sate", V=40, B=3, T=50, seed=8, **sate)
    if os.environ.get("GOLDEN_ONLY", "") != "":
        return
    small_ctc = {k: v for k, v in small.items() if not k.startswith("decoder")}
    ctc_greedy_case("ctc_greedy_transformer", outdir, V=40, B=4, T=64, seed=4, **small_ctc)
    ctc_greedy_case("ctc_greedy_conformer", outdir, V=40, B=4, T=64, seed=5, **small_ctc, **conf)


if __name__ == "__main__":
    if sys.argv[1] == "--check":
        sys.argv.pop(1)
        check(sys.argv[1])
    main()
