"""End-to-end GPU parity of the HIP model path against (i) the golden vectors dumped from the reference and
(ii) the CPU oracle on fresh seeded inputs.

Tolerance (BASELINE.json north_star): logits within 1e-3 relative in fp32; CTC-greedy token ids bit-exact.
bf16 runs are checked against the same fp32 targets with a bf16-sized tolerance (documented per assert)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import s2t_oracle as O  # noqa: E402
from s2t_amd import criterions as C  # noqa: E402
from s2t_amd import pdss2t_transformer as PDS  # noqa: E402
from s2t_amd import s2t_sate as SATE  # noqa: E402
from s2t_amd import s2t_transformer as M  # noqa: E402

DEV = "cuda"
CASES = ["transformer_small", "conformer_small", "conformer_ragged", "pds_small", "pds_conformer_small", "sate_small",
         "conformer_interctc", "conformer_compress", "transformer_compress", "pds_fusion_small",
         "conformer_compress_margin", "transformer_compress_margin", "pds_fusion_learned"]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def args_from_cfg(cfg, vocab):
    a = Namespace(
        input_feat_per_channel=80, input_channels=1, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
        share_decoder_input_output_embed=True, share_ctc_and_embed=True, encoder_embed_norm=True,
        encoder_no_scale_embedding=True, vocab_size=vocab,
    )
    for k, v in cfg.items():
        setattr(a, k, v)
    if "decoder_embed_dim" not in cfg:
        a.decoder_layers = 0
    return a


def build(z, dtype, ctc_only=False):
    cfg = O.cfg_from_golden(z)
    vocab = z["w::decoder.embed_tokens.weight"].shape[0] if "w::decoder.embed_tokens.weight" in z.files \
        else z["w::encoder.ctc.ctc_projection.weight"].shape[0]
    args = args_from_cfg(cfg, vocab)
    task = M.FakeTask(vocab)
    arch = str(cfg.get("arch", ""))
    if ctc_only:
        args.ctc_weight = 1.0
        model = M.S2TCTCModel.build_model(args, task)
    elif arch.startswith("pdss2t"):
        model = PDS.PDSS2TTransformerModel.build_model(args, task)
    elif arch.startswith("s2t_sate"):
        model = SATE.S2TSATEModel.build_model(args, task)
    else:
        model = M.S2TTransformerModel.build_model(args, task)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    missing, unexpected = model.load_state_dict(sd, strict=True), None  # strict: every reference key must exist
    model.prepare(dtype, DEV)
    return model, cfg


def _skip_bf16_compress(name, dtype):
    if name.endswith("_compress") and dtype == torch.bfloat16:
        # which frames survive is a hard threshold on a posterior: these two fixtures put blank posteriors right at it, bf16
        # logits move some across, after which the tensors are no longer comparable frame by frame.  bf16 compression is
        # compared on the *_compress_margin fixtures (every posterior >= 0.16 from the threshold), which are not skipped.
        pytest.skip("threshold-straddling compression fixture: fp32 only (bf16: *_compress_margin)")


def bf16_round_npz(z):
    """The fixture with every floating-point weight and the input features rounded to bf16 (what the bf16 HIP model
    computes on): the oracle evaluated on THESE is the bf16-aware target — the only difference left is activation rounding."""
    out = {}
    for k in z.files:
        v = z[k]
        if (k.startswith("w::") or k == "in::src_tokens") and v.dtype.kind == "f" and "running_" not in k:
            v = torch.from_numpy(v).bfloat16().float().numpy()
        out[k] = v

    class _Z(dict):
        files = list(out.keys())

    return _Z(out)


# measured per-tensor relative L2 gradient error (bf16 HIP model vs the oracle on bf16-rounded weights), worst tensor of each
# fixture on MI355X; the asserted bound is twice this
BF16_GRAD_ERR = {}


def rel_err(got, ref):
    got = got.detach().float().cpu().numpy()
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3)])
def test_eval_forward_matches_reference(golden_dir, name, dtype, tol):
    _skip_bf16_compress(name, dtype)
    z = load(golden_dir, name)
    model, cfg = build(z, dtype)
    model.eval()
    src = torch.from_numpy(z["in::src_tokens"]).to(DEV)
    lens = torch.from_numpy(z["in::src_lengths"]).to(DEV)
    prev = torch.from_numpy(z["in::prev_output_tokens"]).to(DEV)
    with torch.no_grad():
        enc = model.encoder(src, lens)
        logits, _ = model.decoder(prev, encoder_out=enc)
    assert (enc["encoder_padding_mask"][0].cpu().numpy() == z["out::encoder_padding_mask"]).all()
    assert enc["encoder_out"][0].shape == z["out::encoder_out"].shape  # (T', B, d) time-major like the reference
    # fp32: 1e-3 relative (north_star); bf16: 12 layers of bf16 rounding on O(1) activations
    assert rel_err(enc["encoder_out"][0], z["out::encoder_out"]) < tol
    assert rel_err(enc["ctc_logit"][0], z["out::ctc_logit"]) < tol
    assert rel_err(logits, z["out::decoder_logits"]) < tol
    i = 0
    while "out::inter_ctc_logit_%d" % i in z.files:
        assert rel_err(enc["inter_ctc_logits"][i][0], z["out::inter_ctc_logit_%d" % i]) < tol
        if "out::inter_ctc_mask_%d" % i in z.files:
            assert (enc["inter_ctc_logits"][i][1].cpu().numpy() == z["out::inter_ctc_mask_%d" % i]).all()
        i += 1


# bf16 bounds = twice the error measured on MI355X between the bf16 HIP model and the fp32 oracle on the SAME bf16-rounded
# weights and inputs (per fixture: eval forward max-relative error, loss relative error, worst per-tensor gradient
# relative L2).  At d = 32 a single ReLU / GLU gate that rounds across zero moves a whole small tensor, which is why the
# toy fixtures still show 5-27 % on their worst tensor; the same comparison at d = 256 (test_configs_fullsize_gpu.py)
# measures 5.5 % worst / 0.8 % median and is bounded accordingly.
BF16_BOUNDS = {  # measured:              forward  loss     worst gradient tensor
    "transformer_small": (1.4e-2, 2e-4, 0.11),    # 0.0068  0.00004  0.055  encoder.layers.0.ffn.w_1.weight
    "conformer_small": (2.8e-2, 2e-4, 0.14),      # 0.0138  0.00001  0.070  conv_module.pointwise_conv1.weight
    "conformer_ragged": (1.5e-2, 2e-4, 0.27),     # 0.0074  0.00000  0.136  layers.1.conv_module.depthwise_conv.weight
    "pds_small": (1.5e-2, 6e-4, 0.20),            # 0.0076  0.00026  0.101  decoder.layers.1.final_layer_norm.bias
    "pds_conformer_small": (3.0e-2, 2e-4, 0.18),  # 0.0149  0.00001  0.090  decoder.layers.1.final_layer_norm.bias
    "sate_small": (2.2e-2, 1e-3, 0.13),           # 0.0109  0.00046  0.064  textual_encoder.layers.0.fc1.weight
    "conformer_interctc": (2.5e-2, 2e-4, 0.53),   # 0.0124  0.00003  0.267  layers.0.conv_module.depthwise_conv.weight (4 layers, d = 32)
    "pds_fusion_small": (1.9e-2, 3e-4, 0.38),     # 0.0093  0.00014  0.190  stage2.0.ffn_norm.bias
    # CTC-guided compression with every frame's blank posterior >= 0.16 from the threshold (oracle/gen_golden.py searches
    # the seeds): bf16 rounding moves no frame across it, so the compressed tensors compare frame by frame
    "conformer_compress_margin": (2.3e-2, 5e-4, 0.23),    # 0.0113  0.00021  0.115  encoder.layers.1.conv_norm.bias
    "pds_fusion_learned": (2.3e-2, 2e-4, 0.25),           # 0.0114  0.00003  0.122  encoder.stage4.0.ffn_norm.bias (learned fusion weights)
    "transformer_compress_margin": (1.1e-2, 1e-3, 0.10),  # 0.0052  0.00049  0.049  decoder.layers.0.final_layer_norm.bias
}


@pytest.mark.parametrize("name", [c for c in CASES if not c.endswith("_compress")])
def test_bf16_against_oracle_on_rounded_weights(golden_dir, name):
    """bf16 HIP model vs the (reference-pinned) oracle evaluated in fp32 on the bf16-rounded weights and inputs of the
    fixture: eval-mode outputs, the joint loss and every parameter gradient.  Only activation rounding separates the two."""
    z = bf16_round_npz(load(golden_dir, name))
    model, cfg = build(z, torch.bfloat16)
    src = torch.from_numpy(z["in::src_tokens"])
    lens = torch.from_numpy(z["in::src_lengths"])
    prev = torch.from_numpy(z["in::prev_output_tokens"])
    target = torch.from_numpy(z["in::target"])
    fwd_tol, loss_tol, grad_tol = BF16_BOUNDS[name]
    # ---- eval forward
    model.eval()
    W0 = O.weights_from_golden(z)
    with torch.no_grad():
        enc = model.encoder(src.to(DEV), lens.to(DEV))
        logits, _ = model.decoder(prev.to(DEV), encoder_out=enc)
        enc_o = O.ENCODERS[O.encoder_kind(cfg)](src, lens, W0, cfg, training=False)
        logits_o = O.decoder_forward(prev, enc_o, W0, cfg)
    fe = max(rel_err(enc["encoder_out"][0], enc_o["encoder_out"][0].numpy()),
             rel_err(enc["ctc_logit"][0], enc_o["ctc_logit"][0].numpy()), rel_err(logits, logits_o.numpy()))
    # ---- loss and gradients (training mode: BatchNorm batch statistics)
    model.train()
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(model.decoder.output_projection.weight.shape[0]),
                                                       label_smoothing=0.1, ctc_weight=cfg["ctc_weight"],
                                                       inter_ctc_weight=float(cfg.get("inter_ctc_weight", 0.0) or 0.0))
    sample = {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": int(z["in::ntokens"])}
    model.flat.zero_grad()
    loss, _, log = crit(model, sample)
    loss.backward()
    torch.cuda.synchronize()
    W = O.weights_from_golden(z, requires_grad=True)
    loss_o, _ = O.joint_loss(W, cfg, src, lens, prev, target, eps=0.1, training=True, use_torch_ctc=True)
    loss_o.backward()
    le = abs(float(loss.detach()) - float(loss_o.detach())) / abs(float(loss_o.detach()))
    ptr = {k: v.data_ptr() for k, v in model.state_dict().items()}
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        if k.endswith(("k_proj.bias", "linear_k.bias")):
            continue  # mathematically zero gradient: rounding noise on both sides
        if "fusion_downsampling" in k and (k.endswith(("depthwise_conv.bias", "pointwise_conv1.bias")) or
                                           (k.endswith("depthwise_conv.weight") and p.shape[-1] == 1)):
            continue  # shift / 1-tap scale in front of BatchNorm: mathematically zero as well
        gs = [W[k2].grad for k2 in W if k2 in ptr and ptr[k2] == ptr[k] and W[k2].grad is not None]
        if not gs:
            continue
        go = sum(gs)
        if ("subsample" in k or ("downsampling" in k and ".conv." in k)) and go.dim() == 3:
            go = go.permute(0, 2, 1)  # stored [Cout][k][Cin]
        gf = p.grad.detach().float().cpu()
        e = float((gf - go).norm() / go.norm().clamp_min(1e-3))
        if e > worst[1]:
            worst = (k, e)
    print("bf16 vs oracle(rounded) %-20s forward %.4f  loss %.5f  worst gradient %s %.4f" % (name, fe, le, worst[0], worst[1]))
    assert fe < fwd_tol, fe
    assert le < loss_tol, le
    assert worst[1] < grad_tol, worst


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 1e-4, 5e-3)])
def test_loss_and_grads_match_reference(golden_dir, name, dtype, tol, gtol):
    _skip_bf16_compress(name, dtype)
    z = load(golden_dir, name)
    model, cfg = build(z, dtype)
    model.train()
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(model.decoder.output_projection.weight.shape[0]),
                                                       label_smoothing=0.1, ctc_weight=cfg["ctc_weight"],
                                                       inter_ctc_weight=float(cfg.get("inter_ctc_weight", 0.0) or 0.0))
    sample = {
        "net_input": {"src_tokens": torch.from_numpy(z["in::src_tokens"]).to(DEV),
                      "src_lengths": torch.from_numpy(z["in::src_lengths"]).to(DEV),
                      "prev_output_tokens": torch.from_numpy(z["in::prev_output_tokens"]).to(DEV)},
        "target": torch.from_numpy(z["in::target"]).to(DEV),
        "ntokens": int(z["in::ntokens"]),
    }
    model.flat.zero_grad()
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    torch.cuda.synchronize()
    for k in ("loss", "trans_loss", "nll_loss", "ctc_loss", "inter_ctc_loss"):
        if "out::" + k not in z.files:
            continue
        ref = float(z["out::" + k])
        assert abs(log[k] - ref) <= tol * abs(ref), (k, log[k], ref)
    assert int(log["total"]) == int(z["out::total"])
    if dtype == torch.float32:
        assert int(log["n_correct"]) == int(z["out::n_correct"])
    assert sample_size == int(z["out::sample_size"])
    params = dict(model.named_parameters())
    worst = ("", 0.0)
    n = 0
    for k in z.files:
        if not k.startswith("grad::"):
            continue
        key = k[6:]
        ref = z[k]
        g = params[key].grad.detach().float().cpu().numpy()
        if ("subsample" in key or ("downsampling" in key and ".conv." in key)) and ref.ndim == 3:
            g = g.transpose(0, 2, 1)  # stored [Cout][k][Cin]
        if key.endswith("k_proj.bias") or key.endswith("linear_k.bias"):
            # mathematically ZERO gradient (softmax is invariant to a per-row score shift): both sides hold rounding
            # noise only, so compare against the size of the sibling q-bias gradient instead of its own
            sib = z["grad::" + key.replace("k_proj", "q_proj").replace("linear_k", "linear_q")]
            assert np.abs(g - ref).max() < gtol * max(np.abs(sib).max(), 1e-3), key
            continue
        if "fusion_downsampling" in key and (key.endswith(("depthwise_conv.bias", "pointwise_conv1.bias")) or
                                             (key.endswith("depthwise_conv.weight") and ref.shape[-1] == 1)):
            # a per-channel shift (or, for a 1-tap kernel, scale) in front of BatchNorm: mathematically ZERO gradient
            # (up to eps), rounding noise on both sides;
            # compare on the scale of the BatchNorm gain's gradient
            sib = z["grad::" + key.rsplit(".", 2)[0] + ".norm.weight"]
            assert np.abs(g - ref).max() < gtol * max(np.abs(sib).max(), 1e-3), key
            continue
        if dtype == torch.float32:
            err = np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-3)
        else:
            # bf16: a ReLU pre-activation that rounds across 0 flips one derivative -> an isolated O(1) element error;
            # the per-tensor relative L2 error is the meaningful figure (5-8 % measured on these 2-layer models)
            err = np.linalg.norm(g - ref) / max(np.linalg.norm(ref), 1e-3)
        if err > worst[1]:
            worst = (key, err)
        n += 1
    assert n > 20
    # the 4-layer fixture accumulates twice the bf16 rounding of the 2-layer ones (measured 0.27 on a depthwise kernel);
    # so do the three fused branches of the PDS fusion fixture (0.25 on a LayerNorm bias of stage 2)
    assert worst[1] < (2 * gtol if (dtype == torch.bfloat16 and name in ("conformer_interctc", "pds_fusion_small")) else gtol), worst
    # BatchNorm running statistics moved exactly as nn.BatchNorm1d moves them
    bufs = dict(model.named_buffers())
    for k in z.files:
        if k.startswith("bn_after::") and ("running_mean" in k or "running_var" in k):
            got = bufs[k[len("bn_after::"):]].cpu().numpy()
            np.testing.assert_allclose(got, z[k], rtol=50 * tol, atol=50 * tol)


@pytest.mark.parametrize("name", ["ctc_greedy_transformer", "ctc_greedy_conformer"])
def test_ctc_greedy_ids_bit_exact(golden_dir, name):
    z = load(golden_dir, name)
    model, cfg = build(z, torch.float32, ctc_only=True)
    model.eval()
    dec = M.CTCDecoder([model], None, None, blank_idx=0)
    sample = {"net_input": {"src_tokens": torch.from_numpy(z["in::src_tokens"]).to(DEV),
                            "src_lengths": torch.from_numpy(z["in::src_lengths"]).to(DEV)}}
    hyps = dec.generate([model], sample)
    assert [len(h[0]["tokens"]) for h in hyps] == z["out::hyp_lengths"].tolist()
    assert torch.cat([h[0]["tokens"] for h in hyps]).tolist() == z["out::hyp_tokens"].tolist()
    np.testing.assert_allclose(np.array([float(h[0]["score"]) for h in hyps]), z["out::hyp_scores"], rtol=1e-3, atol=1e-3)


def test_state_dict_keys_match_reference(golden_dir):
    """Checkpoint compatibility: same keys and shapes as the reference's state_dict."""
    for name in ("transformer_small", "conformer_small", "pds_small", "sate_small", "conformer_interctc",
                 "conformer_compress", "transformer_compress", "pds_fusion_small"):
        z = load(golden_dir, name)
        model, _ = build(z, torch.float32)
        sd = model.state_dict()
        ref = {k[3:]: z[k].shape for k in z.files if k.startswith("w::")}
        assert set(sd.keys()) == set(ref.keys())
        for k, shp in ref.items():
            assert tuple(sd[k].shape) == tuple(shp), k


@pytest.mark.parametrize("conformer", [False, True])
def test_fresh_inputs_against_oracle(conformer):
    """Seeded random model at a mid size (d=64, 3 layers, ragged batch incl. a 1-frame-short row) vs the CPU oracle."""
    torch.manual_seed(11)
    V = 53
    args = M.recipe_args(conformer=conformer, encoder_embed_dim=64, encoder_ffn_embed_dim=128, encoder_layers=3,
                         decoder_layers=2, decoder_embed_dim=64, decoder_ffn_embed_dim=128, encoder_attention_heads=4,
                         decoder_attention_heads=4, subsampling_filter=96, vocab_size=V)
    model = M.S2TTransformerModel.build_model(args, M.FakeTask(V))
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    model.prepare(torch.float32, DEV)
    model.eval()
    B, T = 5, 131
    lens = torch.tensor([131, 130, 97, 80, 79])
    src = torch.randn(B, T, 80, generator=g)
    for b in range(B):
        src[b, lens[b]:] = 0
    prev = torch.randint(4, V, (B, 7), generator=g)
    prev[:, 0] = 2
    prev[3, 5:] = 1
    W = {k: v.detach().cpu().float() for k, v in model.state_dict().items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    with torch.no_grad():
        enc = model.encoder(src.to(DEV), lens.to(DEV))
        logits, _ = model.decoder(prev.to(DEV), encoder_out=enc)
        enc_o = O.encoder_forward(src, lens, W, cfg, training=False)
        logits_o = O.decoder_forward(prev, enc_o, W, cfg)
    assert rel_err(enc["encoder_out"][0], enc_o["encoder_out"][0].numpy()) < 1e-3
    assert rel_err(enc["ctc_logit"][0], enc_o["ctc_logit"][0].numpy()) < 1e-3
    assert rel_err(logits, logits_o.numpy()) < 1e-3
    hy, _ = O.ctc_greedy(enc_o["ctc_logit"][0], enc_o["encoder_padding_mask"][0])
    model.encoder.ctc_out_dtype = torch.float32
    dec = M.CTCDecoder([model.encoder], None, None)

    class _EncOnly(torch.nn.Module):
        def __init__(self, e):
            super().__init__()
            self.e = e

        def forward(self, src_tokens, src_lengths):
            return self.e(src_tokens, src_lengths)

    dec.model = _EncOnly(model.encoder)
    hyps = dec.generate(None, {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV)}})
    assert [h[0]["tokens"].tolist() for h in hyps] == [h.tolist() for h in hy]


@pytest.mark.parametrize("conformer", [False, True])
def test_bf16_fused_attention_model_vs_oracle_and_composed(conformer, monkeypatch):
    """64-wide heads in bf16 take the fused attention kernels (csrc/attention_fused.hip): loss and gradients against the
    fp32 CPU oracle (bf16-sized tolerance) and, tighter, against the GEMM-composed path on the same bf16 weights."""
    torch.manual_seed(7)
    V = 61
    args = M.recipe_args(conformer=conformer, encoder_embed_dim=128, encoder_ffn_embed_dim=256, encoder_layers=2,
                         decoder_layers=2, decoder_embed_dim=128, decoder_ffn_embed_dim=256, encoder_attention_heads=2,
                         decoder_attention_heads=2, subsampling_filter=96, vocab_size=V)
    model = M.S2TTransformerModel.build_model(args, M.FakeTask(V))
    W = {k: v.detach().clone().float().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = {k: getattr(args, k) for k in vars(args)}
    model.prepare(torch.bfloat16, DEV)
    model.train()
    g = torch.Generator().manual_seed(3)
    B, T = 4, 300
    lens = torch.tensor([300, 290, 211, 150])
    src = torch.randn(B, T, 80, generator=g)
    for b in range(B):
        src[b, lens[b]:] = 0
    U = 9
    target = torch.randint(4, V, (B, U), generator=g)
    target[:, -1] = 2
    target[2, 6:] = 1
    target[2, 5] = 2
    prev = torch.roll(target, 1, 1)
    prev[:, 0] = 2
    prev[2, 6:] = 1
    ntok = int((target != 1).sum())
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    sample = {"net_input": {"src_tokens": src.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": ntok}

    def run():
        model.flat.zero_grad()
        loss, _, log = crit(model, sample)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()}

    loss_f, grads_f = run()
    monkeypatch.setenv("S2T_ATTN_COMPOSED", "1")
    loss_c, grads_c = run()
    monkeypatch.delenv("S2T_ATTN_COMPOSED")
    loss_o, _ = O.joint_loss(W, cfg, src, lens, prev, target, eps=0.1, training=True, use_torch_ctc=True)
    loss_o.backward()
    assert abs(loss_f - float(loss_o.detach())) < 2e-2 * abs(float(loss_o.detach()))
    assert abs(loss_f - loss_c) < 5e-3 * abs(loss_c)
    worst_o, worst_c = ("", 0.0), ("", 0.0)
    # tied tensors (decoder embedding = output projection = CTC projection) are separate leaves on the oracle side
    ptr = {k: v.data_ptr() for k, v in model.state_dict().items()}
    for k, gf in grads_f.items():
        if k.endswith("k_proj.bias") or k.endswith("linear_k.bias"):
            continue  # mathematically zero gradient, rounding noise on both sides
        go = sum(W[k2].grad for k2 in W if ptr[k2] == ptr[k] and W[k2].grad is not None)
        if "subsample" in k and go.dim() == 3:
            go = go.permute(0, 2, 1)
        eo = float((gf - go).norm() / go.norm().clamp_min(1e-3))
        ec = float((gf - grads_c[k]).norm() / grads_c[k].norm().clamp_min(1e-3))
        if eo > worst_o[1]:
            worst_o = (k, eo)
        if ec > worst_c[1]:
            worst_c = (k, ec)
    assert worst_o[1] < 1.5e-1, worst_o
    assert worst_c[1] < 8e-2, worst_c


def test_ctc_decoder_inter_logit_option(golden_dir):
    """--ctc-inter-logit k (s2t_ctc.py:276-284): greedy decoding from an intermediate CTC head with ITS padding mask; on
    the compression fixture the heads have different frame counts.  Checked against the oracle's greedy decode of the
    reference's own intermediate logits."""
    from argparse import Namespace
    z = load(golden_dir, "conformer_compress")
    model, cfg = build(z, torch.float32)
    model.eval()
    sample = {"net_input": {"src_tokens": torch.from_numpy(z["in::src_tokens"]).to(DEV),
                            "src_lengths": torch.from_numpy(z["in::src_lengths"]).to(DEV)}}

    class _Enc(torch.nn.Module):
        def __init__(self, e):
            super().__init__()
            self.e = e

        def forward(self, src_tokens, src_lengths):
            return self.e(src_tokens, src_lengths)

    for k in (1, 2):
        dec = M.CTCDecoder([model], Namespace(ctc_inter_logit=k), None, blank_idx=0)
        dec.model = _Enc(model.encoder)
        hyps = dec.generate(None, sample)
        i = 2 - k  # inter_logits[-k] of two heads
        ref_h, _ = O.ctc_greedy(torch.from_numpy(z["out::inter_ctc_logit_%d" % i]), torch.from_numpy(z["out::inter_ctc_mask_%d" % i]))
        assert [h[0]["tokens"].tolist() for h in hyps] == [t.tolist() for t in ref_h]
    with pytest.raises(NotImplementedError):
        M.CTCDecoder([model], Namespace(ctc_self_ensemble=True), None)


@pytest.mark.parametrize("name", ["conformer_compress", "transformer_compress", "conformer_compress_margin"])
def test_bounded_compression_equals_the_reference_on_the_kept_frames(golden_dir, name):
    """CTC-guided compression with the frame axis kept at its uncompressed bound (the capturable form: device-side lengths, no
    host copy): the reference's outputs on the first T' = max(new lengths) frames, padding behind them, the same decoder
    logits."""
    z = load(golden_dir, name)
    model, cfg = build(z, torch.float32)
    model.eval()
    model.encoder.compression_bounded = True
    src = torch.from_numpy(z["in::src_tokens"]).to(DEV)
    lens = torch.from_numpy(z["in::src_lengths"]).to(DEV)
    prev = torch.from_numpy(z["in::prev_output_tokens"]).to(DEV)
    with torch.no_grad():
        enc = model.encoder(src, lens)
        logits, _ = model.decoder(prev, encoder_out=enc)
    ref_mask = z["out::encoder_padding_mask"]
    Tn = ref_mask.shape[1]
    mask = enc["encoder_padding_mask"][0].cpu().numpy()
    assert mask.shape[1] >= Tn and (mask[:, :Tn] == ref_mask).all() and mask[:, Tn:].all()
    assert rel_err(enc["encoder_out"][0][:Tn], z["out::encoder_out"]) < 1e-3
    # (frames behind T' are padding like any other: the final LayerNorm leaves its bias there, the mask marks them)
    assert rel_err(enc["ctc_logit"][0][:Tn], z["out::ctc_logit"]) < 1e-3
    assert rel_err(logits, z["out::decoder_logits"]) < 1e-3


def test_training_step_with_compression_captures_into_a_hipgraph(golden_dir):
    """Trainer.capture switches the encoder to the bounded form: the step with a compression layer is one hipGraph, and its
    replays follow the eager (exact-form) trajectory from the same state (fixture weights, dropout 0)."""
    from s2t_amd.trainer import Trainer
    z = load(golden_dir, "conformer_compress_margin")
    losses = {}
    for mode in ("eager", "graph"):
        model, cfg = build(z, torch.float32)
        model.train()
        for mod in model.modules():
            if isinstance(getattr(mod, "p", None), float):
                mod.p = 0.0
            for attr in ("dropout", "attention_dropout", "activation_dropout", "dropout_p"):
                if isinstance(getattr(mod, attr, None), float):
                    setattr(mod, attr, 0.0)
        crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(model.decoder.output_projection.weight.shape[0]),
                                                           label_smoothing=0.1, ctc_weight=cfg["ctc_weight"])
        tr = Trainer(model, crit, lr=1e-4, warmup_updates=1)
        sample = {"net_input": {"src_tokens": torch.from_numpy(z["in::src_tokens"]).to(DEV),
                                "src_lengths": torch.from_numpy(z["in::src_lengths"]).to(DEV),
                                "prev_output_tokens": torch.from_numpy(z["in::prev_output_tokens"]).to(DEV)},
                  "target": torch.from_numpy(z["in::target"]).to(DEV), "ntokens": int(z["in::ntokens"])}
        if mode == "graph":
            tr.capture(sample, warmup=0)
            assert model.encoder.compression_bounded
            losses[mode] = [float(tr.replay()[0]) for _ in range(4)]
        else:
            losses[mode] = [float(tr.train_step(sample)[0]) for _ in range(4)]
    assert all(np.isfinite(losses["graph"]))
    np.testing.assert_allclose(losses["graph"], losses["eager"], rtol=5e-3)
