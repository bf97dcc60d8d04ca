"""The d = 256 bf16 fast paths inside the model families that the SMALL fixtures (d = 32) never take them through
("weak" item 2): the row-block projection kernels, the fused feed-forward forward / backward kernels and
the fused attention kernels run inside

  * a SATE model (modules/speech_to_text/adapter.py:189-297, models/speech_to_text/s2t_sate.py:973-1075: 4 acoustic + 2
    textual encoder layers, 2 decoder layers, V = 10 000) and
  * a PDS Conformer (models/speech_to_text/pdss2t_transformer.py:1042-1281: 4 stages, d = 256 everywhere)

with the row thresholds lowered so that every such kernel is the one that runs, against the fp32 CPU oracle evaluated on the
SAME bf16-rounded weights and inputs (only activation rounding separates the two), bounds at twice the measured error;

  * configuration 5a at its literal size (12-layer Conformer + CTC head, 256 x 1000 x 80, bf16): the size-independent
    properties — a batch permutation permutes the greedy ids bit for bit, extra zero padding changes no id;
  * a d = 256 bf16 Transformer with an intermediate-CTC tap (the shape of the advisor's round-2 finding: final_norm absent,
    a tap reading a layer output that the fused LayerNorm-backward hand-over also touches): gradients with the hand-over on
    and off are the same arithmetic.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402
from oracle import s2t_oracle as O  # noqa: E402
from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import functional as Fn  # noqa: E402
from s2t_amd import pdss2t_transformer as PDS  # noqa: E402
from s2t_amd import s2t_sate as SATE  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402

DEV = "cuda"
V = 10000


def _perturb(model, seed):
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


def _sample(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    lens = sorted([T] + [int(torch.randint(int(0.6 * T), T + 1, (1,), generator=g)) for _ in range(B - 1)], reverse=True)
    src = torch.randn(B, T, 80, generator=g)
    for b, l in enumerate(lens):
        src[b, l:] = 0
    src = src.bfloat16().float()
    ul = [int(torch.randint(10, 21, (1,), generator=g)) for _ in range(B)]
    U = max(ul) + 1
    target = torch.full((B, U), 1, dtype=torch.long)
    prev = torch.full((B, U), 1, dtype=torch.long)
    for b, u in enumerate(ul):
        toks = torch.randint(4, V, (u,), generator=g)
        target[b, :u] = toks
        target[b, u] = 2
        prev[b, 0] = 2
        prev[b, 1:u + 1] = toks
    return src, torch.tensor(lens), prev, target, int(sum(ul) + B)


def _bf16_vs_oracle(model, args, B, T, seed):
    """Loss and every parameter gradient of the bf16 HIP model against the oracle on the rounded weights; returns
    (relative loss error, {parameter: relative L2 gradient error})."""
    _perturb(model, seed)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.bfloat16().float())
    W = {k: v.detach().clone().float().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    model.prepare(torch.bfloat16, DEV)
    model.train()
    src, lens, prev, target, ntok = _sample(B, T, seed + 1)
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    sample = {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": ntok}
    old = (Fn._FFN_FUSED_MIN_ROWS, Fn._RB_MIN_ROWS)
    Fn._FFN_FUSED_MIN_ROWS, Fn._RB_MIN_ROWS = 256, 256
    calls = {"ffn": 0, "ffn_bwd": 0, "rb": 0}
    from s2t_amd import kernels as K
    orig = (K.ffn_fused_fwd, K.ffn_fused_bwd, K.rowblock_gemm)

    def count(name, fn):
        def w(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return w

    K.ffn_fused_fwd, K.ffn_fused_bwd, K.rowblock_gemm = count("ffn", orig[0]), count("ffn_bwd", orig[1]), count("rb", orig[2])
    try:
        model.flat.zero_grad()
        loss, _, log = crit(model, sample)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        Fn._FFN_FUSED_MIN_ROWS, Fn._RB_MIN_ROWS = old
        K.ffn_fused_fwd, K.ffn_fused_bwd, K.rowblock_gemm = orig
    loss_o, aux = O.joint_loss(W, cfg, src, lens, prev, target, eps=0.1, training=True, use_torch_ctc=True)
    loss_o.backward()
    lo = float(loss_o.detach())
    ptr = {k: v.data_ptr() for k, v in model.state_dict().items()}
    errs = {}
    for k, p in model.named_parameters():
        if k.endswith(("k_proj.bias", "linear_k.bias")):
            continue  # mathematically zero
        go = sum(W[k2].grad for k2 in W if ptr[k2] == ptr[k] and W[k2].grad is not None)
        if ("subsample" in k or ("downsampling" in k and ".conv." in k)) and go.dim() == 3:
            go = go.permute(0, 2, 1)  # stored [Cout][k][Cin]
        gf = p.grad.detach().float().cpu()
        if go.shape != gf.shape:
            go = go.reshape(gf.shape)
        errs[k] = float((gf - go).norm() / go.norm().clamp_min(1e-6))
    return abs(float(loss.detach()) - lo) / abs(lo), errs, calls


def test_sate_d256_bf16_kernels_against_oracle_on_rounded_weights():
    torch.manual_seed(61)
    args = M.recipe_args(conformer=False, vocab_size=V, arch="s2t_sate", encoder_layers=4, text_encoder_layers=2, decoder_layers=2,
                         acoustic_encoder="transformer", adapter="inter_league", textual_encoder_embed_norm=True,
                         textual_encoder_no_scale_embedding=True, encoder_normalize_before=True, decoder_normalize_before=True)
    model = SATE.S2TSATEModel.build_model(args, M.FakeTask(V))
    le, errs, calls = _bf16_vs_oracle(model, args, B=8, T=400, seed=62)
    worst = max(errs.items(), key=lambda kv: kv[1])
    med = float(np.median(list(errs.values())))
    print("SATE d256 bf16: loss %.5f, worst gradient %s %.4f, median %.4f, kernel calls %s" % (le, worst[0], worst[1], med, calls))
    # the fast paths really ran — the fused feed-forward block in the 4 acoustic AND the 2 textual layers
    assert calls["ffn"] >= 6 and calls["ffn_bwd"] >= 6 and calls["rb"] >= 4, calls
    # measured on MI355X (round 3): loss 0.00008, worst 0.036 (decoder.layers.1.fc1.weight), median 0.0084.  Bounds above the
    # spread of this chaotic figure over equally valid kernel variants (test_configs_fullsize_gpu.py: median x 0.6 ... x 3.5)
    assert le < 4e-4
    assert worst[1] < 1.5e-1, worst
    assert med < 4e-2


def test_pds_conformer_d256_bf16_kernels_against_oracle_on_rounded_weights():
    torch.manual_seed(71)
    args = M.recipe_args(conformer=True, vocab_size=V, arch="pdss2t_transformer_s_8", pds_stages=4, pds_layers="1_1_1_1",
                         pds_ratios="2_2_1_2", pds_fusion=False, pds_embed_dims="256_256_256_256", pds_ds_method="conv",
                         pds_embed_norm=True, pds_position_embed="1_1_1_1", pds_kernel_sizes="5_5_5_5",
                         pds_ffn_ratios="8_8_8_8", pds_attn_heads="4_4_4_4", decoder_layers=2)
    model = PDS.PDSS2TTransformerModel.build_model(args, M.FakeTask(V))
    le, errs, calls = _bf16_vs_oracle(model, args, B=4, T=1000, seed=72)
    worst = max(errs.items(), key=lambda kv: kv[1])
    med = float(np.median(list(errs.values())))
    print("PDS Conformer d256 bf16: loss %.5f, worst gradient %s %.4f, median %.4f, kernel calls %s" % (le, worst[0], worst[1], med, calls))
    assert calls["ffn"] >= 4 and calls["ffn_bwd"] >= 4 and calls["rb"] >= 4, calls
    # measured on MI355X (round 3): loss 0.00002, worst 0.046 (decoder.layers.1.fc1.weight), median 0.0024; bounds as above
    assert le < 4e-4
    assert worst[1] < 1.8e-1, worst
    assert med < 2e-2


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


def test_config5a_literal_batch_256_properties():
    """BASELINE.json configuration 5 in its 5a reading at the literal size: s2t_ctc, 12-layer Conformer + CTC head, 256
    utterances of 1000 frames, bf16 with fp32 CTC logits, CTC greedy (models/speech_to_text/s2t_ctc.py:236-349)."""
    B, T = 256, 1000
    torch.manual_seed(1)
    a = M.recipe_args(conformer=True, vocab_size=V, ctc_weight=1.0)
    model = M.S2TCTCModel.build_model(a, M.FakeTask(V)).prepare(torch.bfloat16, DEV)
    model.encoder.ctc_out_dtype = torch.float32
    model.eval()
    sample = bench.synthetic_batch(B, T, V, 17, torch.device(DEV))[0]
    ni = sample["net_input"]
    ids = _greedy(model, ni["src_tokens"], ni["src_lengths"])
    assert sum(len(x) for x in ids) > 0
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(DEV)
    ids_p = _greedy(model, ni["src_tokens"][perm].contiguous(), ni["src_lengths"][perm].contiguous())
    assert [ids[i] for i in perm.tolist()] == ids_p
    # extra zero padding (utterances ending a few frames before T: see test_fullsize_properties_gpu.py)
    src = ni["src_tokens"].clone()
    lens = ni["src_lengths"].clamp(max=T - 16)
    for b in range(B):
        src[b, int(lens[b]):] = 0
    base = _greedy(model, src, lens)
    padded = torch.zeros(B, T + 40, 80, device=DEV)
    padded[:, :T] = src
    assert base == _greedy(model, padded, lens)


def test_config3_literal_batch_64x2000_properties():
    """BASELINE.json configuration 3 at its literal size: the recipe's 4-stage PDS Conformer (3 layers per stage, ratios 2-2-1-2,
    d = 256; models/speech_to_text/pdss2t_transformer.py:1042-1281), 64 utterances of 2000 frames (stage lengths 1004 / 502 /
    502 / 251), bf16 with fp32 CTC logits, where the CPU oracle is too slow to be the checker (the same model runs against it at
    4 x 2000 in fp32, test_configs_fullsize_gpu.py, and at 4 x 1000 in bf16 above).  Size-independent properties of the reference:
      * utterances are independent in eval mode: permuting the batch permutes encoder output, CTC logits and greedy ids BIT
        FOR BIT (the stages run packed in eval: an utterance then sits at another row offset, its arithmetic does not move);
      * extra zero padding behind every utterance changes no greedy id (one workgroup per fused-FFN row block pinned: bit
        equality across ROW COUNTS is a property of one summation order, test_fullsize_properties_gpu.py);
      * one training pass: finite loss and gradients, every parameter receives a gradient, and the summed eval loss is additive
        over utterances."""
    from s2t_amd import kernels as K

    B, T = 64, 2000
    torch.manual_seed(1)
    a = M.recipe_args(conformer=True, vocab_size=V, arch="pdss2t_transformer_s_8", pds_stages=4, pds_layers="3_3_3_3",
                      pds_ratios="2_2_1_2", pds_fusion=False, pds_embed_dims="256_256_256_256", pds_ds_method="conv",
                      pds_embed_norm=True, pds_position_embed="1_1_1_1", pds_kernel_sizes="5_5_5_5", pds_ffn_ratios="8_8_8_8",
                      pds_attn_heads="4_4_4_4")
    model = PDS.PDSS2TTransformerModel.build_model(a, M.FakeTask(V)).prepare(torch.bfloat16, DEV)
    model.encoder.ctc_out_dtype = torch.float32
    model.eval()
    sample = bench.synthetic_batch(B, T, V, 23, torch.device(DEV))[0]
    ni = sample["net_input"]
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).to(DEV)
    with torch.no_grad():
        e0 = model.encoder(ni["src_tokens"], ni["src_lengths"])
        e1 = model.encoder(ni["src_tokens"][perm].contiguous(), ni["src_lengths"][perm].contiguous())
        assert e0["encoder_out"][0].shape[0] == 251
        assert torch.equal(e0["encoder_out"][0][:, perm], e1["encoder_out"][0])
        assert torch.equal(e0["ctc_logit"][0][:, perm], e1["ctc_logit"][0])
    del e0, e1
    ids = _greedy(model, ni["src_tokens"], ni["src_lengths"])
    assert sum(len(x) for x in ids) > 0
    ids_p = _greedy(model, ni["src_tokens"][perm].contiguous(), ni["src_lengths"][perm].contiguous())
    assert [ids[i] for i in perm.tolist()] == ids_p
    _, old, _ = K.ffn_configure()
    K.ffn_configure(split=1)
    try:
        src = ni["src_tokens"].clone()
        lens = ni["src_lengths"].clamp(max=T - 64)
        for b in range(B):
            src[b, int(lens[b]):] = 0
        base = _greedy(model, src, lens)
        padded = torch.zeros(B, T + 40, 80, device=DEV)
        padded[:, :T] = src
        assert base == _greedy(model, padded, lens)
        del padded, src
    finally:
        K.ffn_configure(split=old)
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    with torch.no_grad():
        full = float(crit(model, sample)[0])
        parts = 0.0
        for i in range(0, B, 16):
            sub = {"net_input": {k: v[i:i + 16].contiguous() for k, v in ni.items()},
                   "target": sample["target"][i:i + 16].contiguous(), "ntokens": 1}
            parts += float(crit(model, sub)[0])
    assert abs(full - parts) <= 2e-3 * abs(full), (full, parts)
    model.train()
    model.flat.zero_grad()
    loss = crit(model, sample)[0]
    loss.backward()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and bool(torch.isfinite(model.flat.grad).all())
    dead = [k for k, p in model.named_parameters() if float(p.grad.abs().max()) == 0.0
            and not k.endswith(("k_proj.bias", "linear_k.bias"))]  # (the key bias cancels in the softmax: mathematically zero)
    assert not dead, dead


def test_transformer_d256_bf16_inter_ctc_tap_layernorm_handover_on_off():
    """S2T_FUSE_LN_DROP hands dropout(dx) of a LayerNorm backward to the block in front.  With an intermediate-CTC head
    tapping a layer output (egs/mustc/asr/conf/inter.yaml; models/speech_to_text/s2t_transformer.py:1881-1946) that output
    has TWO consumers: the hand-over must not drop or double the tap's gradient.  d = 256 Transformer (no final_norm
    inside the layers), bf16."""
    def grads(fuse):
        torch.manual_seed(4)
        Vs = 60
        args = M.recipe_args(conformer=False, encoder_layers=4, decoder_layers=1, vocab_size=Vs, dropout=0.2, attention_dropout=0.2,
                             activation_dropout=0.2, inter_ctc_layers="2,3", share_inter_ctc=True, inter_ctc_weight=0.2,
                             ctc_pae="none")
        model = M.S2TTransformerModel.build_model(args, M.FakeTask(Vs)).prepare(torch.bfloat16, DEV)
        model.train()
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(Vs), label_smoothing=0.1, ctc_weight=0.3, inter_ctc_weight=0.2)
        gg = torch.Generator().manual_seed(1)
        B, T = 3, 123
        src = torch.randn(B, T, 80, generator=gg).to(DEV)
        lens = torch.tensor([123, 99, 61]).to(DEV)
        tgt = torch.randint(4, Vs, (B, 7), generator=gg)
        tgt[:, -1] = 2
        prev = torch.roll(tgt, 1, 1)
        sample = {"net_input": {"src_tokens": src, "src_lengths": lens, "prev_output_tokens": prev.to(DEV)},
                  "target": tgt.to(DEV), "ntokens": 21}
        Fn._FUSE_LN_DROP = fuse
        Fn.DROP_STATS.update(handed_over=0, launched=0)
        Fn.DROPOUT.begin_step(torch.device(DEV))
        Fn.DROPOUT.set_seed(7)
        model.flat.zero_grad()
        loss = crit(model, sample)[0]
        loss.backward()
        torch.cuda.synchronize()
        names = {}
        for k, p in model.named_parameters():
            names[k] = p.grad.detach().float().clone()
        return float(loss), names, dict(Fn.DROP_STATS)

    try:
        l1, g1, st1 = grads(True)
        l0, g0, st0 = grads(False)
    finally:
        Fn._FUSE_LN_DROP = True
    assert st0["handed_over"] == 0 and st1["handed_over"] >= 4, (st0, st1)
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    worst = max(((k, float((g1[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-12))) for k in g0), key=lambda kv: kv[1])
    assert worst[1] < 1e-4, worst  # same arithmetic; parameter-level sums use float atomics
