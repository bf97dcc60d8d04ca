"""SURVEY.md §8f row 2 — the NAST stack (``s2t_ctc --encoder-type sate``: XCTC / intermediate XCTC heads, prediction-aware
encoding with the ground-truth curriculum, cross-layer attention) on the HIP path against fixtures dumped from the
reference (oracle/gen_golden.py: nast_case).  fp32: logits 1e-3 relative, losses 1e-4, greedy ids bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import s2t_oracle as O  # noqa: E402
from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402
from tests.test_model_parity_gpu import args_from_cfg, load, rel_err  # noqa: E402

DEV = "cuda"


def build(z, dtype):
    cfg = O.cfg_from_golden(z)
    vocab = z["w::encoder.textual_encoder.embed_tokens.weight"].shape[0]
    args = args_from_cfg(cfg, vocab)
    model = M.S2TCTCModel.build_model(args, M.FakeTask(vocab))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    model.load_state_dict(sd, strict=True)  # every reference key (incl. the unused s2_norm) must exist
    assert set(model.state_dict().keys()) == set(sd.keys())
    model.prepare(dtype, DEV)
    return model, cfg


def sample_of(z):
    return {"net_input": {"src_tokens": torch.from_numpy(z["in::src_tokens"]).to(DEV),
                          "src_lengths": torch.from_numpy(z["in::src_lengths"]).to(DEV),
                          "prev_output_tokens": torch.from_numpy(z["in::prev_output_tokens"]).to(DEV)},
            "target": torch.from_numpy(z["in::target"]).to(DEV),
            "transcript": {"tokens": torch.from_numpy(z["in::transcript"]).to(DEV)},
            "ntokens": int(z["in::ntokens"])}


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 8e-2)])
def test_nast_eval_forward_and_greedy(golden_dir, dtype, tol):
    z = load(golden_dir, "nast_small")
    model, cfg = build(z, dtype)
    model.eval()
    s = sample_of(z)
    with torch.no_grad():
        enc = model.encoder(s["net_input"]["src_tokens"], s["net_input"]["src_lengths"])
    assert rel_err(enc["encoder_out"][0], z["out::encoder_out"]) < tol
    assert rel_err(enc["ctc_logit"][0], z["out::ctc_logit"]) < tol
    assert rel_err(enc["xctc_logit"][0], z["out::xctc_logit"]) < tol
    for i in range(2):
        assert rel_err(enc["inter_ctc_logits"][i][0], z["out::inter_ctc_logit_%d" % i]) < tol
        il = enc["inter_xctc_logits"][i]
        assert rel_err(il[0] if isinstance(il, list) else il, z["out::inter_xctc_logit_%d" % i]) < tol
    if dtype == torch.float32:  # CTCDecoder decodes xctc_logit when the encoder has one (s2t_ctc.py:262-268)
        hyps = M.CTCDecoder([model], None, None, blank_idx=0).generate([model], s)
        assert [len(h[0]["tokens"]) for h in hyps] == z["out::hyp_lengths"].tolist()
        assert torch.cat([h[0]["tokens"] for h in hyps]).tolist() == z["out::hyp_tokens"].tolist()
        np.testing.assert_allclose(np.array([float(h[0]["score"]) for h in hyps]), z["out::hyp_scores"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("name", ["nast_small", "nast_pae_oracle"])
@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-4, 5e-3), (torch.bfloat16, 3e-2, 3e-1)])
def test_nast_ctc_criterion_loss_and_grads(golden_dir, name, dtype, tol, gtol):
    z = load(golden_dir, name)
    model, cfg = build(z, dtype)
    model.train()
    crit = C.CtcCriterion(None, M.FakeTask(40), ctc_weight=float(cfg["ctc_weight"]),
                          inter_ctc_weight=float(cfg["inter_ctc_weight"]), xctc_weight=float(cfg["xctc_weight"]),
                          inter_xctc_weight=float(cfg["inter_xctc_weight"]))
    crit.train()
    n_s2 = int(z["aux::n_s2_layers"])
    drops = z["aux::drop_self_attn_all"].tolist()
    kw = {"drop_self_attn": drops[-n_s2:], "first_pass_kwargs": {"drop_self_attn": drops[:n_s2]}}
    if "aux::xctc_rand_mask" in z.files:
        kw["pae_oracle_masks"] = {"xctc": torch.from_numpy(z["aux::xctc_rand_mask"]).to(DEV)}
    model.flat.zero_grad()
    loss, _, log = crit(model, sample_of(z), **kw)
    loss.backward()
    torch.cuda.synchronize()
    for k in ("ctc_loss", "inter_ctc_loss", "xctc_loss", "inter_xctc_loss", "loss"):
        ref = float(z["out::" + k])
        assert abs(log[k] - ref) <= tol * abs(ref), (k, log[k], ref)
    if dtype == torch.bfloat16 and name == "nast_pae_oracle":
        return  # the Viterbi alignment (hence which frames are replaced) may differ under bf16 logits: losses only
    params = dict(model.named_parameters())
    worst, n = ("", 0.0), 0
    for k in z.files:
        if not k.startswith("grad::"):
            continue
        key, ref = k[6:], z[k]
        g = params[key].grad.detach().float().cpu().numpy()
        if "subsample" in key and ref.ndim == 3:
            g = g.transpose(0, 2, 1)  # stored [Cout][k][Cin]
        if key.endswith(("k_proj.bias", "linear_k.bias")):
            sib = z["grad::" + key.replace("k_proj", "q_proj").replace("linear_k", "linear_q")]
            assert np.abs(g - ref).max() < gtol * max(np.abs(sib).max(), 1e-2), key
            continue
        if dtype == torch.float32:
            err = np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-3)
        else:
            err = np.linalg.norm(g - ref) / max(np.linalg.norm(ref), 1e-3)
        if err > worst[1]:
            worst = (key, err)
        n += 1
    print("nast %s %s worst gradient %s %.4f" % (name, str(dtype), worst[0], worst[1]))
    assert n > 60 and worst[1] < gtol, worst


def test_nast_training_step_captures_into_a_hipgraph():
    """The NAST recipe's step — two encoder passes, the Viterbi alignment oracle between them (criterions/ctc.py:283-345), PAE
    with ground-truth mixing, four CTC terms — has no host round trip: it captures, and its replays train (the alignment
    states stay on the device, torch_imputer.best_alignment_states)."""
    import bench
    from s2t_amd.trainer import Trainer
    V = 60
    task = M.FakeTask(V)
    nast = dict(encoder_type="sate", text_encoder_layers=3, acoustic_encoder="transformer", adapter="inter_league",
                xctc_weight=1.0, ctc_weight=1.0, share_ctc_and_embed=True, share_xctc_and_embed=True, text_no_pos_emb=True,
                textual_encoder_embed_norm=False, textual_encoder_no_scale_embedding=True, encoder_normalize_before=True,
                share_inter_ctc=True, inter_ctc_weight=1.0, inter_ctc_layers="2", inter_xctc_weight=1.0, inter_xctc_layers="2",
                ctc_pae="inter_league", xctc_pae="inter_league", xctc_cross_attn=True, cross_attn_start_layer=2,
                cross_attn_layer=1, cross_attn_collaboration_mode="serial", cross_attn_league_drop_net=True,
                cross_attn_league_drop_net_prob=0.1, xctc_pae_ground_truth_ratio=0.8, xctc_pae_ground_truth_only_mistake=True,
                pae_oracle_smooth=True, encoder_embed_dim=128, encoder_ffn_embed_dim=256, encoder_attention_heads=4,
                encoder_layers=3, subsampling_filter=256, activation_fn="relu")
    a = M.recipe_args(conformer=True, vocab_size=V, dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, **nast)
    torch.manual_seed(3)
    m = M.S2TCTCModel.build_model(a, task).prepare(torch.bfloat16, DEV)
    m.train()
    crit = C.CtcCriterion(None, task, ctc_weight=1.0, inter_ctc_weight=1.0, xctc_weight=1.0, inter_xctc_weight=1.0)
    crit.train()
    tr = Trainer(m, crit, lr=1e-3, warmup_updates=1)
    sample, _ = bench.synthetic_batch(6, 240, V, 1, DEV)
    sample["transcript"] = {"tokens": sample["target"]}
    first = float(tr.train_step(sample)[0])
    tr.capture(sample)
    losses = [float(tr.replay()[0]) for _ in range(12)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), losses
    assert min(losses[-3:]) < first, (first, losses)
