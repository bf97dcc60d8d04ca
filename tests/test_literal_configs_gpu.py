"""The configurations of BASELINE.json that no test had run AS STATED ("weak" 1 - 2):

  * configuration 2' at FULL DEPTH — the 12-encoder / 6-decoder-layer Conformer the bench times — in bf16 on packed rows
    (18 x 1000 frames) against the fp32 CPU oracle on the same bf16-rounded weights: eval logits, loss and every parameter
    gradient.  (Every earlier oracle comparison in bf16 ran 4 + 2 layers; twelve layers of bf16 drift were bounded by no test.)
  * configuration 4 literal — s2t_sate, 12 acoustic + 6 textual encoder layers, 6 decoder layers, inter_league adapter
    (models/speech_to_text/s2t_sate.py:973-1075, egs/mustc/st/conf/reproduction_sate.yaml), one GPU's share 64 x 1000 — through
    the size-independent properties: batch permutation bit-exact, extra padding changes no greedy id, the loss is additive over
    utterances, one captured update = the eager update.  (Its data-parallel half at world size 2: test_ddp_two_ranks_gpu.py.)
  * configuration 5b literal — the reproduction_nast.yaml stack (12 Conformer acoustic + 12 textual layers with cross-layer
    attention, d = 512 / 8 heads / F = 2048, intermediate (X)CTC heads at 6, 9 with PAE), 256 x 1000, greedy on xctc_logit
    (models/speech_to_text/s2t_ctc.py:174-349) — permutation and padding properties.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402
from oracle import s2t_oracle as O  # noqa: E402
from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import kernels as K  # noqa: E402
from s2t_amd import s2t_sate as SATE  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402
from test_configs_fullsize_gpu import PK_B, PK_T, PK_UHI, PK_ULO, _batch, _PackingSpy, _perturb, _targets  # noqa: E402

DEV = "cuda"
V = 10000
S2T_R6_DECODER_PERM_EXACT = True  # measured on MI355X (round 6): with one workgroup per FFN row block the permuted decoder logits are bit-exact


def test_config2p_full_depth_12_6_bf16_against_oracle_on_rounded_weights():
    """The bench's model at its full depth (12 Conformer encoder layers, 6 decoder layers, d = 256, V = 10 000) in the layout
    the bench runs (packed encoder rows and packed target rows: 18 x 1000 frames, 90 - 120 target tokens), bf16, against the
    fp32 oracle on the SAME bf16-rounded weights and inputs.  Eval forward first (encoder output, CTC logits, decoder logits on
    the frames / target positions), then one training pass (loss, every parameter gradient)."""
    from s2t_amd import rows as Rows

    assert Rows.ENABLED and PK_B * 250 >= Rows.MIN_ENC_ROWS and PK_B * PK_UHI >= Rows.MIN_DEC_ROWS
    torch.manual_seed(251)
    args = M.recipe_args(conformer=True, vocab_size=V)
    assert args.encoder_layers == 12 and args.decoder_layers == 6
    model = M.S2TTransformerModel.build_model(args, M.FakeTask(V))
    _perturb(model, 252)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.bfloat16().float())
    state = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    model.prepare(torch.bfloat16, DEV)
    B, T = PK_B, PK_T
    src, lens, g = _batch(B, T, 253)
    src = src.bfloat16().float()
    target, prev, ul = _targets(B, g, PK_ULO, PK_UHI)

    # ---- eval forward
    model.eval()
    with torch.no_grad(), _PackingSpy() as spy:
        enc = model.encoder(src.to(DEV), lens.to(DEV))
        logits, _ = model.decoder(prev.to(DEV), encoder_out=enc)
        enc_o = O.encoder_forward(src, lens, state, cfg, training=False)
        logits_o = O.decoder_forward(prev, enc_o, state, cfg)
    assert "enc" in spy.tags and "dec" in spy.tags, spy.tags
    olen = [((int(l) - 1) // 2 + 1 - 1) // 2 + 1 for l in lens]
    fm = torch.zeros(enc_o["encoder_out"][0].shape[:2], dtype=torch.bool)  # (T', B)
    for b, n in enumerate(olen):
        fm[:n, b] = True
    tm = prev.ne(1)

    def rel_on(got, ref, m):
        got = got.detach().float().cpu()
        return float((got - ref)[m].abs().max() / ref[m].abs().max().clamp_min(1e-6))

    e1 = rel_on(enc["encoder_out"][0], enc_o["encoder_out"][0], fm)
    e2 = rel_on(enc["ctc_logit"][0], enc_o["ctc_logit"][0], fm)
    e3 = rel_on(logits, logits_o, tm)
    # per-frame arg-max agreement of the CTC logits (what greedy decoding reads), counted on the frames
    am = (enc["ctc_logit"][0].float().cpu().argmax(-1) == enc_o["ctc_logit"][0].argmax(-1))[fm].float().mean()
    print("12/6 bf16 eval vs oracle on rounded weights: encoder_out %.4f ctc_logit %.4f decoder logits %.4f, frame arg-max agreement %.4f"
          % (e1, e2, e3, float(am)))
    # measured on MI355X (round 6): 0.0138 / 0.0083 / 0.0052, arg-max agreement 1.0000 (the same model at 4 + 2 layers: 0.0111 /
    # 0.0069 / 0.0032 — twelve layers cost a quarter more, not three times); bounds = 2 x measured
    assert e1 < 2.8e-2 and e2 < 1.7e-2 and e3 < 1.1e-2, (e1, e2, e3)
    assert float(am) > 0.99, float(am)
    del enc, logits, enc_o, logits_o

    # ---- one training pass
    W = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in state.items()}
    model.train()
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    sample = {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": int(sum(ul) + B)}
    with _PackingSpy() as spy:
        model.flat.zero_grad()
        loss, _, log = crit(model, sample)
        loss.backward()
        torch.cuda.synchronize()
    assert "enc" in spy.tags and "dec" in spy.tags, spy.tags
    loss_o, aux = O.joint_loss(W, cfg, src, lens, prev, target, eps=0.1, training=True, use_torch_ctc=True)
    loss_o.backward()
    lo = float(loss_o.detach())
    le = abs(float(loss.detach()) - lo) / abs(lo)
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
    med = float(np.median(list(errs.values())))
    by_layer = {}
    for k, v in errs.items():
        if k.startswith("encoder.layers."):
            by_layer.setdefault(int(k.split(".")[2]), []).append(v)
    print("12/6 bf16 training pass: loss rel %.5f, gradient relative L2 worst %s %.4f, median %.4f" % (le, worst[0], worst[1], med))
    print("    median per encoder layer (0 = bottom, furthest from the loss): " +
          " ".join("%d:%.4f" % (i, float(np.median(v))) for i, v in sorted(by_layer.items())))
    for k_, v_ in sorted(errs.items(), key=lambda kv: -kv[1])[:10]:
        print("    %.4f %s" % (v_, k_))
    # measured on MI355X (round 6): loss 1e-5; worst tensor 0.120 (layer-4 linear_pos.weight: the relative-position tables and biases
    # and the depthwise-conv weights lead, as at 4 + 2 layers where they read 0.054), median 0.0149; the per-layer medians grow from
    # 0.005 (layer 11, next to the loss) to 0.015 (layer 0): bf16 drift accumulates over depth about linearly, no layer stands out.
    # The figure is chaotic in the summation order of any kernel (test_configs_fullsize_gpu.py: x 0.6 ... x 3.5 between equally valid
    # variants), a wrong kernel reads ~1 on its tensor: bounds at 2 x measured.
    assert le < 1e-3, le
    assert worst[1] < 2.4e-1, worst
    assert med < 3e-2, med
    assert max(float(np.median(v)) for v in by_layer.values()) < 3.5e-2


def _sate_args(**kw):
    return M.recipe_args(conformer=False, vocab_size=V, arch="s2t_sate", text_encoder_layers=6, acoustic_encoder="transformer",
                         adapter="inter_league", textual_encoder_embed_norm=True, textual_encoder_no_scale_embedding=True,
                         encoder_normalize_before=True, decoder_normalize_before=True, **kw)


def test_config4_literal_sate_12_6_6_batch_64x1000_properties():
    B, T = 64, 1000
    torch.manual_seed(1)
    a = _sate_args()
    assert a.encoder_layers == 12 and a.text_encoder_layers == 6 and a.decoder_layers == 6
    model = SATE.S2TSATEModel.build_model(a, M.FakeTask(V)).prepare(torch.bfloat16, DEV)
    model.eval()
    sample = bench.synthetic_batch(B, T, V, 29, torch.device(DEV))[0]
    ni = sample["net_input"]
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(7)).to(DEV)
    # --- a batch permutation permutes textual-encoder output, acoustic CTC logits and decoder logits bit for bit
    with torch.no_grad():
        e0 = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
        l0, _ = model.decoder(prev_output_tokens=ni["prev_output_tokens"], encoder_out=e0)
        e1 = model.encoder(src_tokens=ni["src_tokens"][perm].contiguous(), src_lengths=ni["src_lengths"][perm].contiguous())
        l1, _ = model.decoder(prev_output_tokens=ni["prev_output_tokens"][perm].contiguous(), encoder_out=e1)
    assert e0["encoder_out"][0].shape[0] == 250
    assert torch.equal(e0["encoder_out"][0][:, perm], e1["encoder_out"][0])
    assert torch.equal(e0["ctc_logit"][0][:, perm], e1["ctc_logit"][0])
    tmask = ni["prev_output_tokens"][perm].ne(1)
    assert float(e0["encoder_out"][0].float().abs().max()) > 0
    # The decoder's 3 904 target rows run its fused feed-forward block dealt to EIGHT workgroups per row block; which of them finishes
    # a row — and with it the order in which the eight fp32 partial sums of that row meet — follows the row's place in its block,
    # and a permuted (packed) batch puts an utterance's rows elsewhere.  So the decoder logits of a permuted batch agree to
    # rounding noise in the shipped configuration, and bit for bit with one workgroup per row block (one summation order).
    pa, pb = l0[perm][tmask].float(), l1[tmask].float()
    rel = float((pa - pb).norm() / pa.norm())
    agree = float((pa.argmax(-1) == pb.argmax(-1)).float().mean())
    print("config 4, permuted batch, decoder logits in the shipped configuration: relative L2 %.2e, arg-max agreement %.4f" % (rel, agree))
    # measured on MI355X (round 6): relative L2 6.9e-4, arg-max agreement 1.0000
    assert rel <= 2e-3 and agree >= 0.999, (rel, agree)
    _, old_split, _ = K.ffn_configure()
    K.ffn_configure(split=1)
    try:
        with torch.no_grad():
            f0 = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
            m0, _ = model.decoder(prev_output_tokens=ni["prev_output_tokens"], encoder_out=f0)
            f1 = model.encoder(src_tokens=ni["src_tokens"][perm].contiguous(), src_lengths=ni["src_lengths"][perm].contiguous())
            m1, _ = model.decoder(prev_output_tokens=ni["prev_output_tokens"][perm].contiguous(), encoder_out=f1)
        exact = torch.equal(m0[perm][tmask], m1[tmask])
        print("config 4, permuted batch, one workgroup per FFN row block: decoder logits bit-exact = %s (max diff %.3e)"
              % (exact, float((m0[perm][tmask].float() - m1[tmask].float()).abs().max())))
        assert S2T_R6_DECODER_PERM_EXACT is False or exact
        del f0, f1, m0, m1
    finally:
        K.ffn_configure(split=old_split)
    # --- extra zero padding behind every utterance: textual-encoder output on the frames and the decoder's arg-max tokens.
    # (one workgroup per fused-FFN row block pinned: the two buffers differ in row count — test_fullsize_properties_gpu.py)
    _, old, _ = K.ffn_configure()
    K.ffn_configure(split=1)
    try:
        src = ni["src_tokens"].clone()
        lens = ni["src_lengths"].clamp(max=T - 16)
        for b in range(B):
            src[b, int(lens[b]):] = 0
        padded = torch.zeros(B, T + 40, 80, device=DEV)
        padded[:, :T] = src
        with torch.no_grad():
            ea = model.encoder(src_tokens=src, src_lengths=lens)
            la, _ = model.decoder(prev_output_tokens=ni["prev_output_tokens"], encoder_out=ea)
            eb = model.encoder(src_tokens=padded, src_lengths=lens)
            lb, _ = model.decoder(prev_output_tokens=ni["prev_output_tokens"], encoder_out=eb)
        sub = model.encoder.acoustic_encoder.subsample.get_out_seq_lens_tensor(lens.cpu())
        Ta = ea["encoder_out"][0].shape[0]
        valid = (torch.arange(Ta)[:, None] < sub[None, :]).to(DEV)
        xa, xb = ea["ctc_logit"][0].float()[valid], eb["ctc_logit"][0].float()[:Ta][valid]
        ya, yb = ea["encoder_out"][0].float()[valid], eb["encoder_out"][0].float()[:Ta][valid]
        tm = ni["prev_output_tokens"].ne(1)
        za, zb = la.float()[tm], lb.float()[tm]
        stats = [(float((p_ - q_).norm() / p_.norm()), float((p_.argmax(-1) == q_.argmax(-1)).float().mean()))
                 for p_, q_ in ((xa, xb), (ya, yb), (za, zb))]
        print("config 4, 40 extra padded frames: (relative L2, arg-max agreement) acoustic CTC logits %s, textual encoder output %s, "
              "decoder logits %s" % tuple("(%.2e, %.4f)" % st for st in stats))
        # the two buffers have different row counts (16 000 / 16 640 rows): GEMM tile boundaries and with them fp32 summation orders may
        # move, so what holds is agreement to bf16 rounding noise through 24 layers (test_extra_padding_in_the_default_configuration)
        # measured on MI355X (round 6): all three families BIT-EXACT on the frames / target positions (relative L2 0, agreement 1) with one
        # workgroup per FFN row block; what is asserted leaves room for a GEMM tile boundary that moves with the row count
        for rel, agree in stats:
            assert rel <= 2e-3, stats
        assert stats[0][1] >= 0.999 and stats[2][1] >= 0.999, stats
        del padded, src, ea, eb, la, lb
    finally:
        K.ffn_configure(split=old)
    # --- the summed loss is additive over utterances (eval mode: no dropout)
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    with torch.no_grad():
        full = float(crit(model, sample)[0])
        parts = 0.0
        for i in range(0, B, 16):
            sub_ = {"net_input": {k: v[i:i + 16].contiguous() for k, v in ni.items()},
                    "target": sample["target"][i:i + 16].contiguous(), "ntokens": 1}
            parts += float(crit(model, sub_)[0])
    assert abs(full - parts) <= 2e-3 * abs(full), (full, parts)


def test_config4_literal_one_captured_update_equals_the_eager_update():
    """Configuration 4's per-GPU step (64 x 1000, recipe dropout 0.1) as ONE hipGraph against the same update issued eagerly:
    same seeds, same dropout masks, same kernels — same loss, same gradient norm, and fp32 masters that moved the same way."""
    from s2t_amd import functional as Fn
    from s2t_amd.trainer import Trainer

    B, T = 64, 1000
    drop = dict(dropout=0.1, attention_dropout=0.1, activation_dropout=0.1)
    sample = bench.synthetic_batch(B, T, V, 31, torch.device(DEV))[0]
    out = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(1)
        model = SATE.S2TSATEModel.build_model(_sate_args(**drop), M.FakeTask(V)).prepare(torch.bfloat16, DEV)
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
        tr = Trainer(model, crit, lr=2e-3, warmup_updates=4)
        p0 = model.flat.master.detach().clone()
        Fn.DROPOUT.begin_step(torch.device(DEV))
        Fn.DROPOUT.set_seed(500)
        if mode == "eager":
            for _ in range(3):
                loss, log = tr.train_step(sample)
            gn = float(log["gnorm"])
        else:
            tr.capture(sample, warmup=2)
            loss = tr.replay()[0]
            gn = float(tr.hyper[3])
        torch.cuda.synchronize()
        out[mode] = (float(loss), gn, (model.flat.master.detach() - p0).float().cpu())
        if mode == "graph":
            tr.release()
        del tr, model
        torch.cuda.empty_cache()
    (le, ge, de), (lg, gg, dg) = out["eager"], out["graph"]
    print("config 4 third update: eager loss %.3f gnorm %.4f | captured loss %.3f gnorm %.4f | movement %.3e, difference %.3e"
          % (le, ge, lg, gg, float(de.abs().mean()), float((de - dg).abs().mean())))
    assert np.isfinite(le) and np.isfinite(lg)
    assert abs(le - lg) <= 2e-3 * abs(le), (le, lg)
    assert abs(ge - gg) <= 1e-2 * abs(ge), (ge, gg)
    assert float(de.abs().mean()) > 1e-4
    assert float((de - dg).abs().mean()) <= 0.03 * float(de.abs().mean())


NAST = dict(encoder_type="sate", text_encoder_layers=12, acoustic_encoder="transformer", adapter="inter_league",
            xctc_weight=1.0, ctc_weight=1.0, share_ctc_and_embed=True, share_xctc_and_embed=True, text_no_pos_emb=True,
            textual_encoder_embed_norm=False, textual_encoder_no_scale_embedding=True, encoder_normalize_before=True,
            share_inter_ctc=True, inter_ctc_weight=1.0, inter_ctc_layers="6,9", inter_xctc_weight=1.0,
            inter_xctc_layers="6,9", ctc_pae="inter_league", xctc_pae="inter_league", xctc_cross_attn=True,
            cross_attn_start_layer=4, cross_attn_layer=3, cross_attn_collaboration_mode="serial",
            cross_attn_league_drop_net=True, cross_attn_league_drop_net_prob=0.1, xctc_pae_ground_truth_ratio=0.8,
            xctc_pae_ground_truth_only_mistake=True, pae_oracle_smooth=True, encoder_embed_dim=512,
            encoder_ffn_embed_dim=2048, encoder_attention_heads=8, subsampling_filter=2048, activation_fn="relu")


def test_config5b_literal_nast_12_12_d512_batch_256x1000_properties():
    B, T = 256, 1000
    torch.manual_seed(1)
    a = M.recipe_args(conformer=True, vocab_size=V, **NAST)
    assert a.encoder_layers == 12 and a.text_encoder_layers == 12 and a.encoder_embed_dim == 512
    model = M.S2TCTCModel.build_model(a, M.FakeTask(V)).prepare(torch.bfloat16, DEV)
    model.encoder.xctc_out_dtype = torch.float32  # greedy decodes xctc_logit: fp32 there (bit-exact arg-max)
    model.eval()
    sample = bench.synthetic_batch(B, T, V, 37, torch.device(DEV))[0]
    ni = sample["net_input"]
    dec = M.CTCDecoder([model], None, None, blank_idx=0)

    def greedy(src, lens):
        with torch.no_grad():
            hyps = dec.generate([model], {"net_input": {"src_tokens": src, "src_lengths": lens}})
        return [h[0]["tokens"].tolist() for h in hyps]

    ids = greedy(ni["src_tokens"], ni["src_lengths"])
    assert sum(len(x) for x in ids) > 0
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(9)).to(DEV)
    ids_p = greedy(ni["src_tokens"][perm].contiguous(), ni["src_lengths"][perm].contiguous())
    assert [ids[i] for i in perm.tolist()] == ids_p
    # extra zero padding: the frames' XCTC logits agree to bf16 rounding noise and the greedy frames with them (the two buffers
    # have different row counts, so tile boundaries of the GEMM composition — and with them fp32 summation orders — may move:
    # test_fullsize_properties_gpu.py::test_extra_padding_in_the_default_configuration states the same for configuration 2')
    src = ni["src_tokens"].clone()
    lens = ni["src_lengths"].clamp(max=T - 16)
    for b in range(B):
        src[b, int(lens[b]):] = 0
    padded = torch.zeros(B, T + 40, 80, device=DEV)
    padded[:, :T] = src
    with torch.no_grad():
        xa = model.encoder(src_tokens=src, src_lengths=lens)["xctc_logit"][0].float()
        xb = model.encoder(src_tokens=padded, src_lengths=lens)["xctc_logit"][0].float()
    sub = torch.div(torch.div(lens.cpu() - 1, 2, rounding_mode="floor") + 1 - 1, 2, rounding_mode="floor") + 1
    valid = (torch.arange(xa.shape[0])[:, None] < sub[None, :]).to(DEV)
    va, vb = xa[valid], xb[:xa.shape[0]][valid]
    rel = float((va - vb).norm() / va.norm())
    agree = float((va.argmax(-1) == vb.argmax(-1)).float().mean())
    print("config 5b 256 x 1000, 40 extra padded frames: xctc logits relative L2 %.5f, frame arg-max agreement %.4f" % (rel, agree))
    # measured on MI355X (round 6): relative L2 0.00000, agreement 1.0000
    assert rel <= 2e-3, rel
    assert agree >= 0.999, agree
