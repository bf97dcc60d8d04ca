"""Pin the CPU oracle (oracle/s2t_oracle.py) against vectors dumped from the reference itself
(oracle/gen_golden.py).  CPU-only; this is what makes the oracle trustworthy as the HIP checker."""
import os

import numpy as np
import pytest
import torch

from oracle import s2t_oracle as O

CASES = ["transformer_small", "conformer_small", "conformer_ragged", "pds_small", "pds_conformer_small", "sate_small",
         "conformer_interctc", "conformer_compress", "transformer_compress", "pds_fusion_small",
         "conformer_compress_margin", "transformer_compress_margin", "pds_fusion_learned"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_modules(golden_dir):
    z = _load(golden_dir, "modules")
    W = {k[len("sub::w::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sub::w::")}
    x = torch.from_numpy(z["sub::x"]).transpose(0, 1)  # (T,B,C) -> (B,T,C)
    y, yl = O.conv1d_subsample(x, torch.from_numpy(z["sub::lens"]), W, "")
    _close(y.transpose(0, 1), z["sub::y"])
    assert yl.tolist() == z["sub::ylens"].tolist()
    for T in (7, 12):
        _close(O.rel_pos_table(T, 32)[:, None, :], z[f"relpos::T{T}"], atol=1e-6)
    _close(O.sinusoidal_positions(torch.from_numpy(z["sinpos::mask"]), 32), z["sinpos::out"], atol=1e-6)
    _close(O.sinusoidal_positions(torch.from_numpy(z["sinpos::tokens"]), 32), z["sinpos::tokens_out"], atol=1e-6)
    _close(O.layer_norm(torch.from_numpy(z["ln::x"]), torch.from_numpy(z["ln::w"]), torch.from_numpy(z["ln::b"])),
           z["ln::y"], atol=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_eval_forward(golden_dir, name):
    z = _load(golden_dir, name)
    cfg, W = O.cfg_from_golden(z), O.weights_from_golden(z)
    src, lens = torch.from_numpy(z["in::src_tokens"]), torch.from_numpy(z["in::src_lengths"])
    with torch.no_grad():
        enc = O.ENCODERS[O.encoder_kind(cfg)](src, lens, W, cfg, training=False)
        logits = O.decoder_forward(torch.from_numpy(z["in::prev_output_tokens"]), enc, W, cfg)
    assert (enc["encoder_padding_mask"][0].numpy() == z["out::encoder_padding_mask"]).all()
    _close(enc["encoder_out"][0], z["out::encoder_out"], rtol=1e-4, atol=2e-5)
    _close(enc["ctc_logit"][0], z["out::ctc_logit"], rtol=1e-4, atol=2e-5)
    _close(logits, z["out::decoder_logits"], rtol=1e-4, atol=2e-5)
    i = 0
    while "out::inter_ctc_logit_%d" % i in z.files:  # intermediate CTC heads (inter.yaml)
        _close(enc["inter_ctc_logits"][i], z["out::inter_ctc_logit_%d" % i], rtol=1e-4, atol=2e-5)
        if "out::inter_ctc_mask_%d" % i in z.files:  # with CTC-guided compression (dynamic.yaml) T shrinks on the way
            assert (enc["inter_ctc_padding_masks"][i].numpy() == z["out::inter_ctc_mask_%d" % i]).all()
        i += 1
    if name.endswith("_compress"):
        assert enc["encoder_out"][0].shape[0] < enc["inter_ctc_logits"][0].shape[0]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("torch_ctc", [False, True])
def test_loss_and_grads(golden_dir, name, torch_ctc):
    z = _load(golden_dir, name)
    cfg, W = O.cfg_from_golden(z), O.weights_from_golden(z, requires_grad=True)
    bn = {}
    loss, log = O.joint_loss(
        W, cfg,
        torch.from_numpy(z["in::src_tokens"]), torch.from_numpy(z["in::src_lengths"]),
        torch.from_numpy(z["in::prev_output_tokens"]), torch.from_numpy(z["in::target"]),
        eps=0.1, training=True, use_torch_ctc=torch_ctc, bn_stats=bn,
    )
    assert abs(float(loss.detach()) - float(z["out::loss"])) <= 1e-4 * abs(float(z["out::loss"]))
    assert abs(float(log["trans_loss"]) - float(z["out::trans_loss"])) <= 1e-4 * abs(float(z["out::trans_loss"]))
    assert abs(float(log["nll_loss"]) - float(z["out::nll_loss"])) <= 1e-4 * abs(float(z["out::nll_loss"]))
    assert abs(float(log["ctc_loss"]) - float(z["out::ctc_loss"])) <= 1e-4 * abs(float(z["out::ctc_loss"]))
    if "out::inter_ctc_loss" in z.files:
        assert abs(float(log["inter_ctc_loss"]) - float(z["out::inter_ctc_loss"])) <= 1e-4 * abs(float(z["out::inter_ctc_loss"]))
    assert log["n_correct"] == int(z["out::n_correct"]) and log["total"] == int(z["out::total"])
    loss.backward()
    # the reference ties these (share_decoder_input_output_embed / share_ctc_and_embed; SATE also the text embedding);
    # its named_parameters() reports the shared gradient once, under the first-registered name
    tied = [k for k in ("decoder.embed_tokens.weight", "decoder.output_projection.weight",
                        "encoder.ctc.ctc_projection.weight", "encoder.acoustic_encoder.ctc.ctc_projection.weight",
                        "encoder.textual_encoder.embed_tokens.weight") if k in W]
    n = 0
    for k in z.files:
        if not k.startswith("grad::"):
            continue
        key = k[6:]
        if key in tied:
            g = sum(W[t].grad for t in tied if t in W and W[t].grad is not None)
        else:
            g = W[key].grad
        ref = z[k]
        # k_proj.bias has a mathematically zero gradient (softmax is invariant to a per-row shift):
        # both sides hold rounding noise there, hence the absolute floor
        scale = max(np.abs(ref).max(), 1e-3)
        err = np.abs(g.numpy() - ref).max() / scale
        assert err < 2e-3, (key, err)
        n += 1
    assert n > 20
    # BatchNorm running-stat update (momentum 0.1, unbiased variance), convolution.py:60 -> nn.BatchNorm1d
    for k in z.files:
        if k.startswith("bn_after::") and k.endswith("running_mean"):
            mod = k[len("bn_after::"):-len(".running_mean")]
            mean, var, cnt = bn[mod]
            rm0 = z["w::" + mod + ".running_mean"]
            rv0 = z["w::" + mod + ".running_var"]
            _close(0.9 * rm0 + 0.1 * mean.detach().numpy(), z[k], rtol=1e-4, atol=1e-6)
            _close(0.9 * rv0 + 0.1 * var.detach().numpy() * cnt / (cnt - 1), z["bn_after::" + mod + ".running_var"],
                   rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["ctc_greedy_transformer", "ctc_greedy_conformer"])
def test_ctc_greedy(golden_dir, name):
    z = _load(golden_dir, name)
    cfg, W = O.cfg_from_golden(z), O.weights_from_golden(z)
    with torch.no_grad():
        enc = O.encoder_forward(torch.from_numpy(z["in::src_tokens"]), torch.from_numpy(z["in::src_lengths"]), W, cfg)
    _close(enc["ctc_logit"][0], z["out::ctc_logit"], rtol=1e-4, atol=5e-5)
    hyps, scores = O.ctc_greedy(enc["ctc_logit"][0], enc["encoder_padding_mask"][0])
    assert [len(h) for h in hyps] == z["out::hyp_lengths"].tolist()
    assert torch.cat(hyps).tolist() == z["out::hyp_tokens"].tolist()  # bit-exact token ids
    _close(scores, z["out::hyp_scores"], rtol=1e-4, atol=1e-4)
    # and the oracle's decode applied to the reference's own logits gives the same ids
    hyps2, _ = O.ctc_greedy(torch.from_numpy(z["out::ctc_logit"]), torch.from_numpy(z["out::encoder_padding_mask"]))
    assert torch.cat(hyps2).tolist() == z["out::hyp_tokens"].tolist()


def test_ctc_nll_matches_aten_and_edge_cases():
    g = torch.Generator().manual_seed(0)
    T, B, V = 12, 4, 7
    lp = torch.log_softmax(torch.randn(T, B, V, generator=g), -1)
    tg = [torch.tensor([1, 1, 2]), torch.tensor([3]), torch.tensor([], dtype=torch.long), torch.tensor([4, 5, 6, 4, 5, 6, 1])]
    il = torch.tensor([12, 9, 5, 6])  # last: T < needed (7 labels need >= 7 frames) -> inf -> 0
    mine = O.ctc_nll(lp, tg, il)
    ref = torch.nn.functional.ctc_loss(lp, torch.cat(tg), il, torch.tensor([len(t) for t in tg]), blank=0,
                                       reduction="none", zero_infinity=True)
    np.testing.assert_allclose(mine.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    assert float(mine[3]) == 0.0


def test_label_smoothing_properties():
    """The four properties the reference's own tests/test_label_smoothing.py:62-117 asserts."""
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(2, 5, 11, generator=g)
    tgt = torch.randint(2, 11, (2, 5), generator=g)
    tgt[1, 3:] = 1
    loss, nll = O.label_smoothed_nll(logits, tgt, 0.1)
    ce = torch.nn.functional.cross_entropy(logits.view(-1, 11), tgt.view(-1), ignore_index=1, reduction="sum")
    assert abs(float(nll) - float(ce)) < 1e-5  # nll part == plain CE
    loss0, nll0 = O.label_smoothed_nll(logits, tgt, 0.0)
    assert abs(float(loss0) - float(ce)) < 1e-5  # eps = 0 == CE
    logits2 = logits.clone()
    logits2[1, 3:] = torch.randn(2, 11, generator=g)  # padding invariance
    loss2, _ = O.label_smoothed_nll(logits2, tgt, 0.1)
    assert abs(float(loss2) - float(loss)) < 1e-5
    per = sum(O.label_smoothed_nll(logits[b:b + 1, u:u + 1], tgt[b:b + 1, u:u + 1], 0.1)[0] for b in range(2) for u in range(5))
    assert abs(float(per) - float(loss)) < 1e-4  # reduce == sum of unreduced


def test_utterance_cmvn():
    x = torch.randn(50, 80, generator=torch.Generator().manual_seed(2)) * 3 + 1
    y = O.utterance_cmvn(x)
    assert y.mean(0).abs().max() < 1e-5 and (y.std(0, unbiased=False) - 1).abs().max() < 1e-4


def test_beam_search_matches_reference_generator(golden_dir):
    """the oracle's list-based beam search (no incremental state) against hypotheses produced by
    the reference's SequenceGenerator (incremental decoder) on the same weights."""
    z = np.load(os.path.join(golden_dir, "beam_search_transformer.npz"))
    W = O.weights_from_golden(z)
    cfg = O.cfg_from_golden(z)
    src = torch.from_numpy(z["in::src_tokens"])
    lens = torch.from_numpy(z["in::src_lengths"])
    beam, mlb = int(z["gen::beam"]), int(z["gen::max_len_b"])
    hyps = O.beam_search(src, lens, W, cfg, beam=beam, max_len_b=mlb)
    assert [len(h) for h in hyps] == z["gen::n_hyps"].tolist()
    lengths = set()
    for b, hb in enumerate(hyps):
        for k, h in enumerate(hb):
            assert h["tokens"] == z["out::tokens_%d_%d" % (b, k)].tolist(), (b, k)
            assert abs(h["score"] - float(z["out::score_%d_%d" % (b, k)])) < 1e-4
            np.testing.assert_allclose(np.array(h["positional_scores"]), z["out::pos_scores_%d_%d" % (b, k)], atol=1e-4)
            lengths.add(len(h["tokens"]))
    assert len(lengths) > 2  # the fixture exercises early </s> as well as the forced one at max_len


def test_ctc_prefix_scorer_identities():
    """The prefix scorer restated from ESPnet's published algorithm has no reference-side fixture (third-party, absent);
    it is pinned through identities against ATen's CTC loss: walking a label sequence y token by token and closing with
    </s> gives log p_ctc(y | x) = -ctc_loss(y), and the prefix probabilities of all one-token prefixes plus the
    probability of the empty output sum to one."""
    g = torch.Generator().manual_seed(5)
    T, V, blank, eos = 23, 9, 0, 2
    lp = torch.log_softmax(torch.randn(T, V, generator=g) * 2, -1)
    sc = O.CTCPrefixScore(lp.numpy(), blank, eos)
    for y in ([4], [3, 3, 5], [6, 7, 6, 6, 8, 3], []):
        state, prefix, psi_prev = sc.initial_state(), [eos], 0.0
        for tok in y:
            psi, st = sc(prefix, [tok, 5], state)
            assert psi[0] <= psi_prev + 1e-5  # a longer prefix is never more probable
            state, psi_prev, prefix = st[0], float(psi[0]), prefix + [tok]
        psi, _ = sc(prefix, [eos, blank], state)
        ref = torch.nn.functional.ctc_loss(lp[:, None, :], torch.tensor([y], dtype=torch.long).view(1, -1), torch.tensor([T]),
                                           torch.tensor([len(y)]), blank=blank, reduction="none", zero_infinity=False)
        assert abs(float(psi[0]) + float(ref[0])) < 1e-3, (y, float(psi[0]), float(ref[0]))
        assert psi[1] == np.float32(O.CTCPrefixScore.logzero)
    # sum over first tokens of p(prefix = c) + p(empty output) == 1
    # (index 2 is also an ordinary CTC label: its prefix probability comes from a scorer whose </s> id matches nothing)
    cs = [c for c in range(V) if c != blank]
    psi, _ = O.CTCPrefixScore(lp.numpy(), blank, -1)([eos], cs, sc.initial_state())
    p_first = np.exp(psi.astype(np.float64)).sum()
    p_empty = np.exp(float(sc([eos], [eos], sc.initial_state())[0][0]))
    assert abs(p_first + p_empty - 1.0) < 1e-4


def test_joint_ctc_beam_search_runs_and_differs(golden_dir):
    """ctc_weight > 0 changes scores (and is well formed); ctc_weight == 0 is the plain search pinned above."""
    z = np.load(os.path.join(golden_dir, "beam_search_transformer.npz"))
    W, cfg = O.weights_from_golden(z), O.cfg_from_golden(z)
    src, lens = torch.from_numpy(z["in::src_tokens"])[:1], torch.from_numpy(z["in::src_lengths"])[:1]
    beam, mlb = int(z["gen::beam"]), int(z["gen::max_len_b"])
    plain = O.beam_search(src, lens, W, cfg, beam=beam, max_len_b=mlb)
    joint = O.beam_search(src, lens, W, cfg, beam=beam, max_len_b=mlb, ctc_weight=0.3)
    assert len(joint[0]) >= 1 and all(h["tokens"][-1] == 2 for h in joint[0])
    assert [h["score"] for h in joint[0]] != [h["score"] for h in plain[0]]


# ---- the NAST stack (s2t_ctc --encoder-type sate, XCTC, PAE, cross-layer attention) ---------------
def _nast_inputs(z):
    return (torch.from_numpy(z["in::src_tokens"]), torch.from_numpy(z["in::src_lengths"]),
            torch.from_numpy(z["in::target"]), torch.from_numpy(z["in::transcript"]))


def test_nast_eval_forward_and_greedy(golden_dir):
    z = _load(golden_dir, "nast_small")
    cfg, W = O.cfg_from_golden(z), O.weights_from_golden(z)
    assert O.encoder_kind(cfg) == "sate"
    src, lens, _, _ = _nast_inputs(z)
    with torch.no_grad():
        enc = O.sate_encoder_forward(src, lens, W, cfg, training=False)
    _close(enc["encoder_out"][0], z["out::encoder_out"], rtol=1e-4, atol=2e-5)
    _close(enc["ctc_logit"][0], z["out::ctc_logit"], rtol=1e-4, atol=5e-5)
    _close(enc["xctc_logit"][0], z["out::xctc_logit"], rtol=1e-4, atol=5e-5)
    for key in ("inter_ctc_logit", "inter_xctc_logit"):
        i = 0
        while "out::%s_%d" % (key, i) in z.files:
            _close(enc[key + "s"][i], z["out::%s_%d" % (key, i)], rtol=1e-4, atol=5e-5)
            i += 1
        assert i == 2
    hyps, scores = O.ctc_greedy(enc["xctc_logit"][0], enc["encoder_padding_mask"][0])  # s2t_ctc.py:262-268: xctc first
    assert [len(h) for h in hyps] == z["out::hyp_lengths"].tolist()
    assert torch.cat(hyps).tolist() == z["out::hyp_tokens"].tolist()
    _close(scores, z["out::hyp_scores"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["nast_small", "nast_pae_oracle"])
def test_nast_ctc_criterion_loss_and_grads(golden_dir, name):
    z = _load(golden_dir, name)
    cfg, W = O.cfg_from_golden(z), O.weights_from_golden(z, requires_grad=True)
    src, lens, target, transcript = _nast_inputs(z)
    n_s2 = int(z["aux::n_s2_layers"])
    drops = z["aux::drop_self_attn_all"].tolist()
    masks = {"xctc": torch.from_numpy(z["aux::xctc_rand_mask"])} if "aux::xctc_rand_mask" in z.files else None
    loss, log, enc = O.ctc_criterion_loss(W, cfg, src, lens, target, transcript, training=True, oracle_masks=masks,
                                          drop_self_attn=drops[-n_s2:], drop_self_attn_first=drops[:n_s2])
    for k in ("ctc_loss", "inter_ctc_loss", "xctc_loss", "inter_xctc_loss"):
        assert abs(float(log[k]) - float(z["out::" + k])) <= 2e-4 * abs(float(z["out::" + k])), k
    assert abs(float(loss) - float(z["out::loss"])) <= 2e-4 * abs(float(z["out::loss"]))
    if name == "nast_pae_oracle":
        assert any(drops) and enc["xctc_force_emit"] is not None and bool((enc["xctc_force_emit"] >= 0).any())
    loss.backward()
    tied = [k for k in W if k.endswith(("embed_tokens.weight", "ctc.ctc_projection.weight", "xctc.ctc_projection.weight"))]
    n = 0
    for k in z.files:
        if not k.startswith("grad::"):
            continue
        key = k[6:]
        g = sum(W[t].grad for t in tied if W[t].grad is not None) if key in tied else W[key].grad
        ref = z[k]
        # key-projection biases have a mathematically zero gradient (softmax shift invariance): rounding noise on both
        # sides, proportional to the loss scale (~2000 here)
        floor = 2e-2 if key.endswith(("linear_k.bias", "k_proj.bias")) else 1e-3
        scale = max(np.abs(ref).max(), floor)
        assert g is not None, key
        assert np.abs(g.numpy() - ref).max() / scale < 2e-3, key
        n += 1
    assert n > 60


def test_oracle_trainer_trajectory(golden_dir):
    """Row a22: the oracle's restatement of scale -> clip -> Adam -> inverse-sqrt schedule (trainer.py:714-759,
    utils.py:328-369, optim/adam.py:146-226, inverse_square_root_schedule.py:59-85) against five updates run by the
    reference's own FairseqAdam / clip_grad_norm / InverseSquareRootSchedule (oracle/gen_golden.py: trainer_case)."""
    z = np.load(os.path.join(golden_dir, "trainer_conformer_small.npz"))
    cfg = O.cfg_from_golden(z)
    W = O.weights_from_golden(z, requires_grad=True)
    hp = {k[4:]: z[k] for k in z.files if k.startswith("hp::")}
    src, lens = torch.from_numpy(z["in::src_tokens"]), torch.from_numpy(z["in::src_lengths"])
    prev, target = torch.from_numpy(z["in::prev_output_tokens"]), torch.from_numpy(z["in::target"])
    n = len(z["out::loss"])
    losses, gnorms, lrs = O.train_trajectory(
        W, cfg, src, lens, prev, target, int(z["in::ntokens"]), n, float(hp["lr"]), tuple(float(b) for b in hp["betas"]),
        float(hp["eps"]), float(hp["weight_decay"]), float(hp["clip_norm"]), int(hp["warmup_updates"]),
        float(hp["warmup_init_lr"]))
    np.testing.assert_allclose(lrs, z["out::lr"], rtol=1e-12)
    np.testing.assert_allclose(losses, z["out::loss"], rtol=2e-4)
    np.testing.assert_allclose(gnorms, z["out::gnorm"], rtol=2e-3)
    for k in z.files:
        # the key bias gradient is mathematically zero (softmax is shift invariant): what reaches Adam is rounding noise,
        # which Adam normalises to steps of +-lr — neither implementation's value means anything
        if k.startswith("after::") and z[k].dtype.kind == "f" and not k.endswith(("linear_k.bias", "k_proj.bias")):
            got = W[k[7:]].detach().numpy()
            assert np.abs(got - z[k]).max() <= 2e-3 * max(np.abs(z[k]).max(), 1e-3), k
