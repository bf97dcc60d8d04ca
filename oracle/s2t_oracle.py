"""CPU oracle for the S2T hot path — TEST INFRASTRUCTURE, NOT A PRODUCT PATH.

A plain fp32 PyTorch-CPU restatement, written from the maths, of the reference's algorithm for the
path named in BASELINE.json.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the checker.  The shipped
package ``s2t_amd`` never imports it and has no CPU fallback.

Pinning: every function here is checked against golden vectors dumped from the reference itself
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``) by ``tests/test_oracle_golden.py`` — EXCEPT ``kaldi_fbank``
(row a1): the reference delegates it to torchaudio.compliance.kaldi.fbank, a third-party dependency that is neither in
/root/reference nor in this image, so that function restates torchaudio's published algorithm and its PARITY IS UNPINNED
against the reference.  Since round 6 it is cross-checked against an INDEPENDENT restatement of the same call — Hugging Face
transformers' Speech2TextFeatureExtractor (the port of this fairseq front-end; its numpy path for machines without torchaudio),
run here by oracle/gen_golden_fbank.py into tests/golden/fbank_hf_speech2text.npz: agreement to 1e-6 absolute on log-mel values
(tests/test_frontend.py).  Two third-party-independent restatements agreeing is evidence, not a pin to the reference.

All tensors are batch-major ``(B, T, C)`` inside the oracle; the reference is time-major between
modules (``(T, B, C)``), so boundary outputs are transposed back where the reference returns them.
Weights are addressed by the reference's own ``state_dict`` keys.

Reference file:line citations are relative to ``/root/reference/fairseq``.
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

NEG_INF = float("-inf")


# ----------------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------------
def lengths_to_padding_mask(lens: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """data/data_utils.py:518-522 — True where t >= len."""
    max_len = int(lens.max()) if max_len is None else max_len
    return torch.arange(max_len)[None, :] >= lens[:, None]


def subsampled_lengths(lens: torch.Tensor, n_layers: int = 2) -> torch.Tensor:
    """modules/speech_to_text/subsampling.py:153-154 — l -> floor((l-1)/2)+1 per strided conv."""
    for _ in range(n_layers):
        lens = torch.div(lens - 1, 2, rounding_mode="floor") + 1
    return lens


def layer_norm(x, w, b, eps=1e-5):
    """modules/layer_norm.py:30-35 -> torch.nn.LayerNorm(eps=1e-5): biased variance over the last dim."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def activation(name: str, x):
    """modules/activations.py — relu / swish (x * sigmoid(x)) / gelu."""
    if name == "relu":
        return torch.clamp_min(x, 0.0)
    if name == "swish":
        return x * torch.sigmoid(x)
    if name == "gelu":
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    raise ValueError(name)


def glu_channels(x):
    """GLU over the channel (last) dim: first half * sigmoid(second half)."""
    a, g = x.chunk(2, dim=-1)
    return a * torch.sigmoid(g)


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


# ----------------------------------------------------------------------------------------------
# positional encodings
# ----------------------------------------------------------------------------------------------
def sinusoidal_table(n_pos: int, dim: int, padding_idx: int = 1) -> torch.Tensor:
    """modules/sinusoidal_positional_embedding.py:36-58 — [sin | cos] halves (not interleaved),
    log-timescale step log(1e4)/(dim/2-1), row ``padding_idx`` zeroed."""
    half = dim // 2
    step = math.log(10000.0) / (half - 1)
    inv = torch.exp(torch.arange(half, dtype=torch.float32) * -step)
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(n_pos, 1)], dim=1)
    tab[padding_idx] = 0.0
    return tab


def sinusoidal_positions(tokens_or_mask: torch.Tensor, dim: int, padding_idx: int = 1) -> torch.Tensor:
    """modules/sinusoidal_positional_embedding.py:60-105 + utils.py:240-250 (make_positions).

    ``tokens_or_mask`` is (B, T) — token ids for the decoder, or the *bool padding mask* for the S2T
    encoder (s2t_transformer.py:1785), in which case True(=1) equals ``padding_idx`` and marks padding.
    Non-pad positions count 1,2,... and are offset by ``padding_idx`` (first real frame -> row 2)."""
    x = tokens_or_mask.long()
    nonpad = (x != padding_idx).long()
    pos = torch.cumsum(nonpad, dim=1) * nonpad + padding_idx
    tab = sinusoidal_table(padding_idx + 1 + x.size(1), dim, padding_idx)
    return tab[pos]  # (B, T, dim)


def rel_pos_table(T: int, dim: int) -> torch.Tensor:
    """modules/positional_encoding.py:121-166 — (2T-1, dim); row n encodes relative offset T-1-n,
    interleaved sin (even cols) / cos (odd cols)."""
    rel = torch.arange(T - 1, -T, -1, dtype=torch.float32)[:, None]
    inv = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    tab = torch.zeros(2 * T - 1, dim)
    tab[:, 0::2] = torch.sin(rel * inv)
    tab[:, 1::2] = torch.cos(rel * inv)
    return tab


# ----------------------------------------------------------------------------------------------
# subsampler
# ----------------------------------------------------------------------------------------------
def conv1d_subsample(x, lens, W: Dict[str, torch.Tensor], prefix: str, n_layers: int = 2):
    """modules/speech_to_text/subsampling.py:106-159 — n x [Conv1d(k, stride 2, pad k//2) -> GLU(ch)].

    x (B, T, C) -> (B, T', C_out).  Padded input frames are whatever the caller put there (zeros from
    the collater); no masking happens between the convolutions."""
    y = x.transpose(1, 2)  # (B, C, T)
    for i in range(n_layers):
        w = W[f"{prefix}layers.{i}.0.weight"]
        b = W[f"{prefix}layers.{i}.0.bias"]
        k = w.size(2)
        y = F.conv1d(y, w, b, stride=2, padding=(k - 1) // 2)
        a, g = y.chunk(2, dim=1)
        y = a * torch.sigmoid(g)
    return y.transpose(1, 2), subsampled_lengths(lens, n_layers)


# ----------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------
def _split_heads(x, h):
    B, T, d = x.shape
    return x.view(B, T, h, d // h).transpose(1, 2)  # (B, h, T, dk)


def mha(q_in, kv_in, W, prefix, h, key_padding_mask=None, causal=False):
    """modules/multihead_attention.py:161-431 — q scaled by dk^-0.5 (:265), scores + causal mask,
    key-pad -> -inf (:383-395), fp32 softmax (:403), @V, out_proj."""
    d = q_in.size(-1)
    dk = d // h
    q = linear(q_in, W[prefix + "q_proj.weight"], W[prefix + "q_proj.bias"]) * dk**-0.5
    k = linear(kv_in, W[prefix + "k_proj.weight"], W[prefix + "k_proj.bias"])
    v = linear(kv_in, W[prefix + "v_proj.weight"], W[prefix + "v_proj.bias"])
    q, k, v = _split_heads(q, h), _split_heads(k, h), _split_heads(v, h)
    s = q @ k.transpose(-1, -2)  # (B,h,Tq,Tk)
    if causal:
        Tq, Tk = s.shape[-2:]
        s = s + torch.triu(torch.full((Tq, Tk), NEG_INF), diagonal=1)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], NEG_INF)
    p = torch.softmax(s.float(), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(q_in.size(0), q_in.size(1), d)
    return linear(o, W[prefix + "out_proj.weight"], W[prefix + "out_proj.bias"])


def rel_pos_mha(x, pos_tab, W, prefix, h, key_padding_mask=None):
    """modules/espnet_multihead_attention.py:313-356 (+ :88-155, :292-311).

    scores[i,j] = ((q_i+u)·k_j + (q_i+v)·p[T-1-i+j]) / sqrt(dk) where p = linear_pos(pos_tab) and
    row n of pos_tab encodes offset T-1-n, so the pair (i,j) sees the encoding of offset i-j — this is
    what the reference's pad-reshape-slice ``rel_shift`` selects.  Key-pad -> -inf, clamp to +-1e8
    (:127-128; -inf becomes -1e8), fp32 softmax, @V, linear_out."""
    B, T, d = x.shape
    dk = d // h
    q = _split_heads(linear(x, W[prefix + "linear_q.weight"], W[prefix + "linear_q.bias"]), h)
    k = _split_heads(linear(x, W[prefix + "linear_k.weight"], W[prefix + "linear_k.bias"]), h)
    v = _split_heads(linear(x, W[prefix + "linear_v.weight"], W[prefix + "linear_v.bias"]), h)
    p = linear(pos_tab, W[prefix + "linear_pos.weight"]).view(2 * T - 1, h, dk).transpose(0, 1)  # (h,2T-1,dk)
    u = W[prefix + "pos_bias_u"][None, :, None, :]
    vb = W[prefix + "pos_bias_v"][None, :, None, :]
    ac = (q + u) @ k.transpose(-1, -2)  # (B,h,T,T)
    bd_full = (q + vb) @ p.transpose(-1, -2)[None]  # (B,h,T,2T-1)
    idx = (T - 1) - torch.arange(T)[:, None] + torch.arange(T)[None, :]  # (T,T) in [0, 2T-2]
    bd = torch.gather(bd_full, 3, idx[None, None].expand(B, h, T, T))
    s = (ac + bd) / math.sqrt(dk)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], NEG_INF)
    s = s.clamp(min=-1e8, max=1e8)
    pr = torch.softmax(s.float(), dim=-1)
    o = (pr @ v).transpose(1, 2).reshape(B, T, d)
    return linear(o, W[prefix + "linear_out.weight"], W[prefix + "linear_out.bias"])


# ----------------------------------------------------------------------------------------------
# Conformer convolution module
# ----------------------------------------------------------------------------------------------
def conv_module(x, pad_mask, W, prefix, act: str, training: bool, bn_eps=1e-5):
    """modules/convolution.py:76-120 — mask-zero; pointwise d->2d (no bias); GLU; depthwise K (pad K//2,
    no bias); BatchNorm1d over (B,T) *including padded frames*; activation; pointwise d->d; mask-zero.

    Returns (y, (batch_mean, batch_var_biased, n)) so tests can check the running-stat update."""
    if pad_mask is not None:
        x = x.masked_fill(pad_mask[:, :, None], 0.0)
    y = linear(x, W[prefix + "pointwise_conv1.weight"][:, :, 0])
    y = glu_channels(y)
    wd = W[prefix + "depthwise_conv.weight"]  # (d,1,K)
    K = wd.size(2)
    y = F.conv1d(y.transpose(1, 2), wd, None, padding=(K - 1) // 2, groups=wd.size(0)).transpose(1, 2)
    stats = None
    if training:
        mean = y.mean(dim=(0, 1))
        var = ((y - mean) ** 2).mean(dim=(0, 1))
        stats = (mean, var, y.size(0) * y.size(1))
    else:
        mean = W[prefix + "norm.running_mean"]
        var = W[prefix + "norm.running_var"]
    y = (y - mean) / torch.sqrt(var + bn_eps) * W[prefix + "norm.weight"] + W[prefix + "norm.bias"]
    y = activation(act, y)
    y = linear(y, W[prefix + "pointwise_conv2.weight"][:, :, 0])
    if pad_mask is not None:
        y = y.masked_fill(pad_mask[:, :, None], 0.0)
    return y, stats


def ffn(x, W, prefix, act, n1="w_1", n2="w_2"):
    """modules/s2t_transformer_layer.py:55-66 — Linear -> act -> Linear."""
    hdn = activation(act, linear(x, W[f"{prefix}{n1}.weight"], W[f"{prefix}{n1}.bias"]))
    return linear(hdn, W[f"{prefix}{n2}.weight"], W[f"{prefix}{n2}.bias"])


# ----------------------------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------------------------
def encoder_layer(x, pad_mask, pos_tab, W, prefix, cfg, training, bn_stats=None):
    """modules/s2t_transformer_layer.py:229-322, pre-LN (encoder_normalize_before=True)."""
    h = cfg["encoder_attention_heads"]
    act = cfg["encoder_activation_fn"]
    if cfg["macaron_style"]:
        y = layer_norm(x, W[prefix + "macaron_norm.weight"], W[prefix + "macaron_norm.bias"])
        x = x + 0.5 * ffn(y, W, prefix + "macaron_ffn.", act)
        scale = 0.5
    else:
        scale = 1.0
    y = layer_norm(x, W[prefix + "self_attn_layer_norm.weight"], W[prefix + "self_attn_layer_norm.bias"])
    if cfg["encoder_attention_type"] == "rel_pos":
        y = rel_pos_mha(y, pos_tab, W, prefix + "self_attn.", h, pad_mask)
    else:
        y = mha(y, y, W, prefix + "self_attn.", h, pad_mask)
    x = x + y
    if cfg["use_cnn_module"]:
        y = layer_norm(x, W[prefix + "conv_norm.weight"], W[prefix + "conv_norm.bias"])
        # conv-module activation is --activation-fn (s2t_transformer_layer.py:125), not the FFN's
        y, st = conv_module(y, pad_mask, W, prefix + "conv_module.", cfg["activation_fn"], training)
        if bn_stats is not None:
            bn_stats[prefix + "conv_module.norm"] = st
        x = x + y
    y = layer_norm(x, W[prefix + "ffn_norm.weight"], W[prefix + "ffn_norm.bias"])
    x = x + scale * ffn(y, W, prefix + "ffn.", act)
    if cfg["use_cnn_module"]:
        x = layer_norm(x, W[prefix + "final_norm.weight"], W[prefix + "final_norm.bias"])
    return x


def inter_ctc_layer_list(cfg):
    """--inter-ctc-layers "6,9" -> [6, 9] (1-based: the head reads the output of that layer; values <= 0 count from the
    top, s2t_transformer.py:1004-1013)."""
    spec = cfg.get("inter_ctc_layers", None)
    if spec is None or str(spec) in ("", "None"):
        return []
    out = []
    for t in str(spec).split(","):
        L = int(t)
        out.append(L + int(cfg["encoder_layers"]) if L <= 0 else L)
    return out


def pae_inter_league(x, logit, w_embed, temperature=1.0, oracle=None, oracle_mask=None, oracle_smooth=False):
    """modules/speech_to_text/adapter.py:189-297, ``inter_league``: x + dist @ W_embed with dist = softmax(logit / tau)
    (:213-217); frames under ``oracle_mask`` take the one-hot distribution of the oracle label instead (:245-262; with
    ``oracle_smooth`` 0.9 + 0.1/V on the label and 0.1/V elsewhere).  x (B,T,d), logit (B,T,V), oracle/mask (B,T)."""
    dist = torch.softmax(logit / float(temperature), dim=-1)
    if oracle is not None:
        V = logit.size(-1)
        od = F.one_hot(oracle, V).to(dist.dtype)
        if oracle_smooth:
            od = torch.where(od == 1, 0.9 + 0.1 / V, 0.1 / V).to(dist.dtype)
        dist = torch.where(oracle_mask[..., None], od, dist)
    return x + dist @ w_embed


def pae_oracle_mask(entry, gt_ratio, adaptive=False, only_mistake=False, mask=None):
    """s2t_transformer.py:1904-1935 / s2t_sate.py:774-797: which frames are fed the ground-truth label.
    entry = (oracle (B,T), best_aligns_pad (B,T), mistake_flag (B,T), mistake_ratio (B,)).  ``mask`` overrides the
    ``torch.rand(...) < prob`` draw (fixtures record the reference's draw).  -> (oracle, mask, force_emit)"""
    oracle, aligns, mistake_flag, mistake_ratio = entry
    if mask is None:
        prob = gt_ratio * mistake_ratio.unsqueeze(-1) if adaptive else gt_ratio
        mask = torch.rand(oracle.size()) < prob
    mask = mask.bool().clone()
    if only_mistake:
        mask = mask & mistake_flag
    return oracle, mask, aligns.masked_fill(~mask, -1)


def _ints_csv(spec):
    return [] if spec is None or str(spec) in ("", "None", "none") else [int(t) for t in str(spec).split(",")]


def encoder_forward(src_tokens, src_lengths, W, cfg, training=False, prefix="encoder.", bn_stats=None,
                    ctc_alignment_oracle=None, oracle_masks=None):
    """models/speech_to_text/s2t_transformer.py:1714-2154 (no mixup; intermediate CTC heads and prediction-aware encoding
    ``ctc_pae inter_league`` included).

    Returns the reference's dict (time-major tensors in lists)."""
    d = cfg["encoder_embed_dim"]
    x, lens = conv1d_subsample(src_tokens, src_lengths, W, prefix + "subsample.")
    T = x.size(1)
    pad_mask = lengths_to_padding_mask(lens, T)
    x = x * (~pad_mask)[:, :, None].to(x.dtype)  # :1765
    if cfg.get("encoder_embed_norm", True):
        x = layer_norm(x, W[prefix + "embed_ln.weight"], W[prefix + "embed_ln.bias"])  # :1769
    if not cfg.get("encoder_no_scale_embedding", True):
        x = x * math.sqrt(d)
    pos_tab = None
    if cfg["encoder_attention_type"] == "rel_pos":
        pos_tab = rel_pos_table(T, d)  # not added to x (:1777-1778)
    else:
        x = x + sinusoidal_positions(pad_mask, d, padding_idx=1)  # :1785-1787
    any_valid = not bool(pad_mask.all())
    inter_layers = inter_ctc_layer_list(cfg)
    inter_logits, inter_masks = [], []
    comp_layers = _ints_csv(cfg.get("compression_layers"))
    comp_thr = [float(t) for t in str(cfg.get("compression_threshold", "1.0")).split(",")]
    comp_thr = comp_thr if len(comp_thr) == len(comp_layers) else comp_thr * len(comp_layers)
    ctc_force_emit = ctc_orc = None
    for i in range(cfg["encoder_layers"]):
        if cfg.get("layer_padding_mask", False) and any_valid:
            x = x.masked_fill(pad_mask[:, :, None], 0.0)  # :1828-1836
        x = encoder_layer(x, pad_mask, pos_tab, W, f"{prefix}layers.{i}.", cfg, training, bn_stats)
        if (i + 1) in inter_layers:
            # intermediate CTC (:1881-1946, ctc_pae none): own LayerNorm ctc_norm{L} (or the final one when
            # share_inter_ctc_norm), then the shared top projection (share_inter_ctc) or the layer's own head
            L = i + 1
            npre = prefix + ("layer_norm." if cfg.get("share_inter_ctc_norm", False) else "ctc_norm%d." % L)
            hpre = prefix + ("ctc." if cfg.get("share_inter_ctc", False) else "inter_ctc%d." % L)
            nx = layer_norm(x, W[npre + "weight"], W[npre + "bias"])
            logit = linear(nx, W[hpre + "ctc_projection.weight"], W[hpre + "ctc_projection.bias"])
            inter_logits.append(logit.transpose(0, 1))
            inter_masks.append(pad_mask)  # the reference's [logit, encoder_padding_mask] pair (:1903)
            if cfg.get("ctc_pae", "none") == "inter_league":  # :1937-1944
                ppre = prefix + ("pae." if cfg.get("share_inter_ctc", False) else "pae%d." % L)
                orc = msk = None
                gt = float(cfg.get("ctc_pae_ground_truth_ratio", 0) or 0)
                if gt > 0 and ctc_alignment_oracle is not None and ctc_alignment_oracle.get("ctc") is not None:
                    if ctc_force_emit is None:
                        ctc_orc = pae_oracle_mask(ctc_alignment_oracle["ctc"], gt,
                                                  cfg.get("xctc_pae_ground_truth_ratio_adaptive", False),
                                                  cfg.get("xctc_pae_ground_truth_only_mistake", False),
                                                  (oracle_masks or {}).get("ctc"))
                        ctc_force_emit = ctc_orc[2]
                    orc, msk = ctc_orc[0], ctc_orc[1]
                x = pae_inter_league(x if cfg.get("pae_unnorm_input", False) else nx, logit,
                                     W[ppre + "embed_adapter.weight"], cfg.get("pae_ctc_temperature", 1.0), orc, msk, False)
            elif cfg.get("ctc_pae", "none") != "none":
                raise NotImplementedError(cfg["ctc_pae"])
            if L in comp_layers:
                # CTC-guided compression (:1948-2040, ``threshold`` metric): drop the frames whose blank posterior reaches
                # the layer's threshold; ``create`` left-packs the kept frames of every utterance (new T = longest),
                # ``mask`` only extends the padding mask; then [compression_norm{L}], [positions for the new layout],
                # padded frames -> 0
                if cfg.get("compression_metric", "ratio") != "threshold":
                    raise NotImplementedError(cfg.get("compression_metric"))
                thr = comp_thr[comp_layers.index(L)]
                blank_prob = torch.softmax(logit, dim=-1)[:, :, 0]
                keep = (blank_prob < thr) & ~pad_mask
                cnt = keep.sum(1)
                if cfg.get("compression_mode", "create") == "create":
                    if int(cnt.min()) > 0 and not bool(keep.all()):
                        Tn = int(cnt.max())
                        out_x = x.new_zeros(x.size(0), Tn, x.size(2))
                        for b in range(x.size(0)):
                            out_x[b, : int(cnt[b])] = x[b][keep[b]]
                        x = out_x
                        pad_mask = lengths_to_padding_mask(cnt, Tn)
                elif cfg["compression_mode"] == "mask":
                    pad_mask = pad_mask | ~keep
                else:
                    raise NotImplementedError(cfg["compression_mode"])
                if cfg.get("compression_norm", False):
                    x = layer_norm(x, W[prefix + "compression_norm%d.weight" % L], W[prefix + "compression_norm%d.bias" % L])
                if cfg.get("compression_pos", False) and cfg["encoder_attention_type"] != "rel_pos":
                    x = x + sinusoidal_positions(pad_mask, d, padding_idx=1)
                x = x.masked_fill(pad_mask[:, :, None], 0.0)
                any_valid = not bool(pad_mask.all())
        if pos_tab is not None and pos_tab.size(0) != 2 * x.size(1) - 1:
            pos_tab = rel_pos_table(x.size(1), d)  # :1838-1843 / :2021-2026: relative positions follow the new length
    x = layer_norm(x, W[prefix + "layer_norm.weight"], W[prefix + "layer_norm.bias"])
    out = {
        "encoder_out": [x.transpose(0, 1)],
        "encoder_padding_mask": [pad_mask],
        "ctc_logit": [],
        "inter_ctc_logits": inter_logits,
        "inter_ctc_padding_masks": inter_masks,
        "ctc_force_emit": ctc_force_emit,
    }
    if prefix + "ctc.ctc_projection.weight" in W:
        logit = linear(x, W[prefix + "ctc.ctc_projection.weight"], W[prefix + "ctc.ctc_projection.bias"])
        out["ctc_logit"] = [logit.transpose(0, 1)]  # (T', B, V)
    return out


# ----------------------------------------------------------------------------------------------
# decoder
# ----------------------------------------------------------------------------------------------
def decoder_forward(prev_output_tokens, enc_out, W, cfg, prefix="decoder.", pad_idx=1):
    """models/transformer.py:1249-1448 + modules/transformer_layer.py:395-543 (pre-LN, teacher forced)."""
    d = cfg["decoder_embed_dim"]
    h = cfg["decoder_attention_heads"]
    emb = W[prefix + "embed_tokens.weight"]
    x = math.sqrt(d) * emb[prev_output_tokens]  # :1314
    x = x + sinusoidal_positions(prev_output_tokens, d, padding_idx=pad_idx)  # :1304,1323
    self_pad = prev_output_tokens.eq(pad_idx)
    self_pad = self_pad if bool(self_pad.any()) else None  # :1340-1342
    mem = enc_out["encoder_out"][0].transpose(0, 1)  # (B, T', d)
    mem_pad = enc_out["encoder_padding_mask"][0]
    for i in range(cfg["decoder_layers"]):
        p = f"{prefix}layers.{i}."
        y = layer_norm(x, W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"])
        x = x + mha(y, y, W, p + "self_attn.", h, self_pad, causal=True)
        y = layer_norm(x, W[p + "encoder_attn_layer_norm.weight"], W[p + "encoder_attn_layer_norm.bias"])
        x = x + mha(y, mem, W, p + "encoder_attn.", h, mem_pad)
        y = layer_norm(x, W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"])
        x = x + ffn(y, W, p, cfg.get("activation_fn", "relu"), n1="fc1", n2="fc2")
    x = layer_norm(x, W[prefix + "layer_norm.weight"], W[prefix + "layer_norm.bias"])
    wout = W.get(prefix + "output_projection.weight", emb)
    return x @ wout.t()  # (B, U, V)


# ----------------------------------------------------------------------------------------------
# beam search
# ----------------------------------------------------------------------------------------------
class CTCPrefixScore:
    """Restatement of the scorer the reference imports at fairseq/sequence_generator.py:17 and drives at :255-388 —
    ``espnet.nets.ctc_prefix_score.CTCPrefixScore`` (ESPnet is an unpinned third-party dependency, setup.py:205, absent
    from the reference tree; algorithm: Watanabe et al. 2017, eq. 51-54; numpy float32, ``logzero = -1e10``).

    x: (T, V) float32 log-probabilities of ONE utterance.  State of a prefix: r (T, 2) — log-probability that frames
    0..t emit the prefix and end in its last label (column 0) / in blank (column 1).  PARITY UNPINNED against ESPnet
    itself; pinned instead by the identity psi(y + </s>) == -CTC-NLL(y) against ATen's ctc_loss (tests)."""

    logzero = -10000000000.0

    def __init__(self, x, blank, eos):
        self.x = np.asarray(x, dtype=np.float32)
        self.blank, self.eos = blank, eos
        self.input_length = len(self.x)

    def initial_state(self):
        r = np.full((self.input_length, 2), self.logzero, dtype=np.float32)
        r[:, 1] = np.cumsum(self.x[:, self.blank], dtype=np.float32)
        return r

    def __call__(self, y, cs, r_prev):
        """y: prefix INCLUDING the leading </s>-as-<sos>; cs: candidate ids.  -> (log_psi (K,), states (K, T, 2))"""
        cs = np.asarray(cs, dtype=np.int64)
        T, K = self.input_length, len(cs)
        out_len = len(y) - 1
        r = np.full((T, 2, K), self.logzero, dtype=np.float32)  # rows below out_len-1 are dead (never read again)
        xs = self.x[:, cs]
        if out_len == 0:
            r[0, 0] = xs[0]
        r_sum = np.logaddexp(r_prev[:, 0], r_prev[:, 1])
        log_phi = np.repeat(r_sum[:, None], K, axis=1)
        if out_len > 0:
            log_phi[:, cs == int(y[-1])] = r_prev[:, 1:2]
        start = max(out_len, 1)
        log_psi = r[start - 1, 0].copy()
        for t in range(start, T):
            r[t, 0] = np.logaddexp(r[t - 1, 0], log_phi[t - 1]) + xs[t]
            r[t, 1] = np.logaddexp(r[t - 1, 0], r[t - 1, 1]) + self.x[t, self.blank]
            log_psi = np.logaddexp(log_psi, log_phi[t - 1] + xs[t])
        log_psi[cs == self.eos] = r_sum[-1]
        log_psi[cs == self.blank] = self.logzero
        return log_psi.astype(np.float32), np.moveaxis(r, 2, 0)


CTC_SCORING_RATIO = 1.5  # sequence_generator.py:19


def beam_search(src_tokens, src_lengths, W, cfg, beam, max_len_a=0.0, max_len_b=200, min_len=1, len_penalty=1.0,
                unk_penalty=0.0, normalize_scores=True, pad=1, eos=2, unk=3, blank=0, max_decoder_positions=1024, ctc_weight=0.0):
    """fairseq/sequence_generator.py:191-614 (_generate) + search.py:101-150 (BeamSearch.step) + :650-786
    (finalize_hypos / is_finished), restated one sentence at a time with plain Python lists and NO incremental state
    (every step re-runs the teacher-forced decoder on the whole prefix):
      * step 0 expands the first beam only; later steps rank all (beam, token) pairs by cumulative log-probability;
      * pad and blank (= <s>, index 0) are never selected, unk is penalised, </s> is forbidden before min_len and
        forced at max_len = min(int(a * src_len + b), max_decoder_positions - 1);
      * of the best 2*beam candidates, an </s> among the FIRST beam is finalised with score / (step+1)^len_penalty
        (while fewer than beam hypotheses are finished); the first beam non-</s> candidates continue;
      * a sentence stops once beam hypotheses are finished or step == max_len; hypotheses are sorted by score.
    ``ctc_weight > 0`` (sequence_generator.py:255-271,355-388): while step <= T', the ``int(1.5 * beam)`` best non-blank
    tokens of every hypothesis get ``(1-w) * lprob + w * (psi(prefix + token) - psi(prefix))`` with psi from
    CTCPrefixScore on the encoder's CTC log-probabilities; the other tokens keep their plain lprob (as in the reference).
    The reference scores every sentence against utterance 0's CTC output (``ctc_lprobs[0]``, i.e. it is a batch-size-1
    path); here each sentence uses its own unpadded frames, which is the same thing at batch size 1.
    Returns, per sentence, a list of dicts {tokens, score, positional_scores}."""
    enc = encoder_forward(src_tokens, src_lengths, W, cfg, training=False)
    B, src_len = src_tokens.shape[:2]
    max_len = min(int(max_len_a * src_len + max_len_b), max_decoder_positions - 1)
    results = []
    for b in range(B):
        enc_b = {"encoder_out": [enc["encoder_out"][0][:, b:b + 1]],
                 "encoder_padding_mask": [enc["encoder_padding_mask"][0][b:b + 1]]}
        hyps = [([eos], [])]  # (tokens incl. the leading </s>, cumulative scores per position)
        finished = []
        if ctc_weight > 0:
            Tb = int((~enc["encoder_padding_mask"][0][b]).sum())
            ctc_lp = torch.log_softmax(enc["ctc_logit"][0][:Tb, b].float(), -1).numpy()
            scorer = CTCPrefixScore(ctc_lp, blank, eos)
            ctc_beam = min(ctc_lp.shape[-1], int(beam * CTC_SCORING_RATIO))
            ctc_state = {(eos,): (scorer.initial_state(), 0.0)}  # prefix -> (state, psi of the prefix)
        for step in range(max_len + 1):
            cands = []
            for bi, (toks, cum) in enumerate(hyps):
                logits = decoder_forward(torch.tensor([toks]), enc_b, W, cfg)[0, -1]
                lp = torch.log_softmax(logits.float(), -1)
                if ctc_weight > 0 and step <= Tb:
                    masked = lp.clone()
                    masked[blank] = NEG_INF
                    ids = torch.topk(masked, ctc_beam).indices.numpy()
                    key = tuple(toks)
                    if key not in ctc_state:  # continued with a token outside its parent's ctc_beam (:363-374)
                        st, _ = ctc_state[key[:-1]]
                        ps, rs = scorer(list(key[:-1]), [key[-1]], st)
                        ctc_state[key] = (rs[0], float(ps[0]))
                    st, prev = ctc_state[key]
                    psi, states = scorer(toks, ids, st)
                    lp[ids] = (1 - ctc_weight) * lp[ids] + ctc_weight * torch.from_numpy(psi - np.float32(prev))
                    for j, v in enumerate(ids.tolist()):
                        ctc_state[key + (v,)] = (states[j], float(psi[j]))
                lp[lp != lp] = NEG_INF
                lp[pad] = NEG_INF
                lp[blank] = NEG_INF
                lp[unk] -= unk_penalty
                if step >= max_len:
                    lp[:eos] = NEG_INF
                    lp[eos + 1:] = NEG_INF
                elif step < min_len:
                    lp[eos] = NEG_INF
                base = cum[-1] if cum else 0.0
                for v in range(lp.numel()):
                    cands.append((float(lp[v]) + base, bi, v))
                if step == 0:
                    break  # all beams are identical at the first step
            cands.sort(key=lambda c: -c[0])
            cands = cands[:min(2 * beam, len(cands) - 1)]
            for sc, bi, v in cands[:beam]:
                if v == eos and sc != NEG_INF and len(finished) < beam:
                    toks, cum = hyps[bi]
                    cumall = cum + [sc]
                    pos = [cumall[0]] + [cumall[i] - cumall[i - 1] for i in range(1, len(cumall))]
                    finished.append({"tokens": toks[1:] + [eos],
                                     "score": sc / (step + 1) ** len_penalty if normalize_scores else sc,
                                     "positional_scores": pos})
            if len(finished) == beam or step == max_len:
                break
            nxt = [(hyps[bi][0] + [v], hyps[bi][1] + [sc]) for sc, bi, v in cands if not (v == eos and sc != NEG_INF)]
            hyps = nxt[:beam]
        finished.sort(key=lambda h: -h["score"])
        results.append(finished)
    return results


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def label_smoothed_nll(logits, target, eps, pad_idx=1):
    """criterions/label_smoothed_cross_entropy.py:42-60, summed over non-pad tokens (fp32 log-softmax)."""
    lp = torch.log_softmax(logits.float(), dim=-1)
    nll = -lp.gather(-1, target[..., None]).squeeze(-1)
    smooth = -lp.sum(-1)
    keep = target.ne(pad_idx)
    nll = (nll * keep).sum()
    smooth = (smooth * keep).sum()
    eps_i = eps / (lp.size(-1) - 1)
    return (1.0 - eps - eps_i) * nll + eps_i * smooth, nll


def ce_accuracy(logits, target, pad_idx=1):
    """criterions/label_smoothed_cross_entropy.py compute_accuracy: argmax == target over non-pad."""
    keep = target.ne(pad_idx)
    n_correct = (logits.argmax(-1).eq(target) & keep).sum()
    return int(n_correct), int(keep.sum())


def ctc_nll(log_probs, targets: List[torch.Tensor], input_lengths, blank=0, zero_infinity=True):
    """CTC negative log-likelihood per utterance, log-space alpha recursion (Graves 2006), matching
    torch.nn.CTCLoss(blank, reduction="none", zero_infinity) as used at criterions/ctc.py:243-245.

    log_probs (T, B, V) fp32 log-softmax; targets: list of B 1-D label tensors (no blanks)."""
    T, B, V = log_probs.shape
    out = []
    for b in range(B):
        y = targets[b]
        S = int(y.numel())
        Tb = int(input_lengths[b])
        L = 2 * S + 1
        ext = torch.full((L,), blank, dtype=torch.long)
        ext[1::2] = y
        lp = log_probs[:Tb, b, :][:, ext]  # (Tb, L)
        # transition s-2 -> s allowed when ext[s] != blank and ext[s] != ext[s-2]
        skip = torch.zeros(L, dtype=torch.bool)
        if L > 2:
            skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        # "log zero" is a large finite negative so that autograd through the recursion stays NaN-free
        LOG0 = -1e30
        alpha = torch.full((L,), LOG0)
        alpha = torch.cat([lp[0, :min(2, L)], alpha[min(2, L):]])
        for t in range(1, Tb):
            a1 = torch.cat([torch.full((1,), LOG0), alpha[:-1]])
            a2 = torch.cat([torch.full((2,), LOG0), alpha[:-2]])[:L]
            a2 = torch.where(skip, a2, torch.full_like(a2, LOG0))
            alpha = torch.logsumexp(torch.stack([alpha, a1, a2]), 0) + lp[t]
        if L > 1:
            ll = torch.logsumexp(torch.stack([alpha[-1], alpha[-2]]), 0)
        else:
            ll = alpha[-1]
        nll = -ll
        if float(nll.detach()) > 1e29:  # no feasible alignment (T too short): inf -> 0 with zero gradient
            nll = nll * 0.0 if zero_infinity else nll + float("inf")
        out.append(nll)
    return torch.stack(out)


def ctc_targets(target, pad_idx=1, eos_idx=2):
    """criterions/ctc.py:516-540 — strip pad and eos; per-utterance label lists."""
    return [row[(row != pad_idx) & (row != eos_idx)] for row in target]


def joint_loss(W, cfg, src_tokens, src_lengths, prev_output_tokens, target, eps=0.1, training=True,
               use_torch_ctc=False, bn_stats=None):
    """criterions/label_smoothed_cross_entropy_with_ctc.py:74-165: CE(label-smoothed) + ctc_weight * CTC."""
    enc = ENCODERS[encoder_kind(cfg)](src_tokens, src_lengths, W, cfg, training=training, bn_stats=bn_stats)
    logits = decoder_forward(prev_output_tokens, enc, W, cfg)
    ce, nll = label_smoothed_nll(logits, target, eps)
    lens = (~enc.get("ctc_padding_mask", enc["encoder_padding_mask"])[0]).long().sum(-1)
    lp = torch.log_softmax(enc["ctc_logit"][0].float(), dim=-1)
    tg = ctc_targets(target)
    if use_torch_ctc:
        flat = torch.cat(tg)
        tl = torch.tensor([len(t) for t in tg])
        ctc = F.ctc_loss(lp, flat, lens, tl, blank=0, reduction="none", zero_infinity=True).sum()
    else:
        ctc = ctc_nll(lp, tg, lens).sum()
    loss = ce + cfg["ctc_weight"] * ctc
    inter = None
    iw = float(cfg.get("inter_ctc_weight", 0.0) or 0.0)
    if iw > 0 and len(enc.get("inter_ctc_logits", [])) > 0:
        # criterions/ctc.py:568-633: the same targets for every intermediate head, losses averaged over the heads
        inter = 0.0
        for j, lg in enumerate(enc["inter_ctc_logits"]):
            ilp = torch.log_softmax(lg.float(), dim=-1)
            # every entry is scored with the lengths of ITS padding mask (criterions/ctc.py:580-590: logit[1])
            il = (~enc["inter_ctc_padding_masks"][j]).long().sum(-1) if "inter_ctc_padding_masks" in enc else lens
            if use_torch_ctc:
                inter = inter + F.ctc_loss(ilp, flat, il, tl, blank=0, reduction="none", zero_infinity=True).sum()
            else:
                inter = inter + ctc_nll(ilp, tg, il).sum()
        inter = inter / len(enc["inter_ctc_logits"])
        loss = loss + iw * inter
    n_correct, total = ce_accuracy(logits, target)
    return loss, {"trans_loss": ce, "nll_loss": nll, "ctc_loss": ctc, "inter_ctc_loss": inter, "n_correct": n_correct,
                  "total": total, "logits": logits, "enc": enc}


# ----------------------------------------------------------------------------------------------
# CTC greedy decode
# ----------------------------------------------------------------------------------------------
def ctc_greedy(ctc_logit_tbv, pad_mask, blank=0):
    """models/speech_to_text/s2t_ctc.py:312-347 — fp32 log-softmax, arg-max (ties -> lowest index),
    padded frames -> blank, collapse repeats, drop blanks.  score = -sum of top-1 log-probs over frames
    whose *unmasked* arg-max is not blank (:327-328 use the index before padding is applied)."""
    lp = torch.log_softmax(ctc_logit_tbv.transpose(0, 1).float(), dim=-1)  # (B,T,V)
    top_p, top_i = lp.max(-1)
    # torch.max returns the first maximal index on CPU, same as topk(1)
    real_i = top_i.masked_fill(pad_mask, blank)
    real_p = top_p.masked_fill(top_i == blank, float(blank))
    scores = -real_p.sum(-1)
    hyps = []
    for b in range(real_i.size(0)):
        seq = real_i[b]
        keep = torch.ones_like(seq, dtype=torch.bool)
        keep[1:] = seq[1:] != seq[:-1]
        h = seq[keep]
        hyps.append(h[h != blank])
    return hyps, scores


# ----------------------------------------------------------------------------------------------
# utterance CMVN (dataloader stage, a2)
# ----------------------------------------------------------------------------------------------
def utterance_cmvn(x, norm_means=True, norm_vars=True):
    """data/audio/feature_transforms/utterance_cmvn.py:31-45 — per-utterance over time:
    x - mean; / sqrt(max(E[x^2]-mean^2, 1e-10)) (variance computed from the *un-centred* moments)."""
    mean = x.mean(0)
    sq = (x**2).sum(0)
    y = x - mean if norm_means else x
    if norm_vars:
        var = sq / x.size(0) - mean**2
        y = y / torch.sqrt(torch.clamp_min(var, 1e-10))
    return y


# ----------------------------------------------------------------------------------------------
# SpecAugment masking (dataloader stage)
# ----------------------------------------------------------------------------------------------
def resize_rows_linear(x, new_rows):
    """cv2.resize(x, dsize=(x.shape[1], new_rows), interpolation=cv2.INTER_LINEAR) for a float32 (rows, cols) array whose
    column count does not change — PARITY UNPINNED: OpenCV is a third-party dependency (unpinned in the reference's
    setup.py, absent from /root/reference and from this image), this restates its published algorithm
    (modules/imgproc/src/resize.cpp, resizeGeneric_ / VResizeLinear): output row dy reads the source coordinate
    fy = (float)((dy + 0.5) * rows / new_rows - 0.5), rows sy = floor(fy) and sy + 1 clamped into [0, rows), blended in
    fp32 as src[sy] * (1 - frac) + src[sy + 1] * frac.  With equal column counts the horizontal pass is the identity."""
    import numpy as np
    rows = x.shape[0]
    dy = np.arange(new_rows, dtype=np.float64)
    fy = ((dy + 0.5) * (float(rows) / float(new_rows)) - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int64)
    frac = (fy - sy.astype(np.float32)).astype(np.float32)
    s0 = np.clip(sy, 0, rows - 1)
    s1 = np.clip(sy + 1, 0, rows - 1)
    x = x.astype(np.float32)
    return (x[s0] * (np.float32(1.0) - frac)[:, None]).astype(np.float32) + (x[s1] * frac[:, None]).astype(np.float32)


def spec_augment(x, freq_mask_n, freq_mask_f, time_mask_n, time_mask_t, time_mask_p, mask_value, rng=None, time_warp_w=0):
    """data/audio/feature_transforms/specaugment.py:79-131: numpy array (T, C) in, augmented copy out.
    time_warp_w > 0 and 2 * time_warp_w < T: split point w0 in [W, T - W) and shift w in [-W + 1, W) are drawn first, the
    first w0 frames are resized to w0 + w and the rest to T - w0 - w frames (resize_rows_linear; pinned against the
    reference's own fixture only for time_warp_w = 0).
    Draw order per mask: width in [0, max), then start in [0, size - width); frequency masks first; the fill value is
    ``mask_value`` or the mean of the un-masked, un-warped spectrogram when None; no time masks when
    min(time_mask_T, floor(T * time_mask_p)) < 1; nothing at all when T == 0 or C < freq_mask_F."""
    import numpy as np
    rng = np.random if rng is None else rng
    y = x.copy()
    T, C = x.shape
    val = x.mean() if mask_value is None else mask_value
    if T == 0 or C < freq_mask_f:
        return x
    if time_warp_w > 0 and 2 * time_warp_w < T:
        w0 = rng.randint(time_warp_w, T - time_warp_w)
        w = rng.randint(-time_warp_w + 1, time_warp_w)
        y = np.concatenate((resize_rows_linear(y[:w0], w0 + w), resize_rows_linear(y[w0:], T - w0 - w)), axis=0)
    for _ in range(freq_mask_n):
        f = rng.randint(0, freq_mask_f)
        f0 = rng.randint(0, C - f)
        if f != 0:
            y[:, f0:f0 + f] = val
    max_t = min(time_mask_t, math.floor(T * time_mask_p))
    if max_t < 1:
        return y
    for _ in range(time_mask_n):
        t = rng.randint(0, max_t)
        t0 = rng.randint(0, T - t)
        if t != 0:
            y[t0:t0 + t, :] = val
    return y


# ----------------------------------------------------------------------------------------------
# Kaldi-compatible log-mel filterbank (dataloader stage, a1) — PARITY UNPINNED
# ----------------------------------------------------------------------------------------------
def kaldi_mel_banks(num_bins, padded_window_size, sample_freq, low_freq=20.0, high_freq=0.0):
    """torchaudio.compliance.kaldi.get_mel_banks (torchaudio is a third-party dependency absent from /root/reference and
    from this image; this restates its published algorithm, no VTLN): triangular filters equally spaced on
    mel(f) = 1127 ln(1 + f/700) between low_freq and nyquist + high_freq; returns (num_bins, padded/2 + 1) with the
    Nyquist column zero (torchaudio pads one zero column)."""
    import numpy as np
    num_fft_bins = padded_window_size // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded_window_size
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    m = mel(fft_bin_width * np.arange(num_fft_bins, dtype=np.float64))[None, :]
    up = (m - left) / (center - left)
    down = (right - m) / (right - center)
    banks = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([banks, np.zeros((num_bins, 1))], 1)


def povey_window(n):
    """torchaudio.compliance.kaldi._feature_window_function('povey'): hann(n, periodic=False) ** 0.85."""
    import numpy as np
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / (n - 1))) ** 0.85


def kaldi_fbank(wave, sample_rate=16000, num_mel_bins=80, frame_length_ms=25.0, frame_shift_ms=10.0, preemph=0.97,
                low_freq=20.0, high_freq=0.0):
    """data/audio/audio_utils.py:59-79 -> torchaudio.compliance.kaldi.fbank(waveform (int16 range), num_mel_bins=80,
    sample_frequency=sr) with torchaudio's defaults: snip_edges, dither 0, remove_dc_offset, pre-emphasis 0.97 (first
    sample against itself), povey window, zero-pad to the next power of two, power spectrum, mel banks (20 Hz ..
    Nyquist), log(max(., float32 eps)).  float64 numpy; wave: 1-D array.  Returns (T, num_mel_bins)."""
    import numpy as np
    wave = np.asarray(wave, dtype=np.float64)
    win = int(sample_rate * frame_length_ms * 0.001)
    shift = int(sample_rate * frame_shift_ms * 0.001)
    padded = 1 << (win - 1).bit_length()
    if wave.shape[0] < win:
        return np.zeros((0, num_mel_bins))
    T = 1 + (wave.shape[0] - win) // shift
    idx = np.arange(T)[:, None] * shift + np.arange(win)[None, :]
    fr = wave[idx]
    fr = fr - fr.mean(1, keepdims=True)
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], 1)
    fr = fr - preemph * prev
    fr = fr * povey_window(win)[None, :]
    spec = np.fft.rfft(fr, n=padded, axis=1)
    power = spec.real ** 2 + spec.imag ** 2
    mel = power @ kaldi_mel_banks(num_mel_bins, padded, float(sample_rate), low_freq, high_freq).T
    return np.log(np.maximum(mel, np.finfo(np.float32).eps))


# ----------------------------------------------------------------------------------------------
# PDS encoder (progressive down-sampling), a16
# ----------------------------------------------------------------------------------------------
def _ints(v):
    return [int(t) for t in str(v).split("_")]


def pds_fusion_stages(cfg):
    """pdss2t_transformer.py:357-391,588-600: the 0-based stages whose outputs are fused (method ``all``; ``conv2``
    transform); fewer than two flagged stages switch fusion off."""
    if not cfg.get("pds_fusion", False):
        return []
    method = str(cfg.get("pds_fusion_method", "none"))
    if method in ("none", "None", ""):
        return []
    kind, _, transform = method.partition("_")
    if kind != "all" or (transform or "conv") != "conv2":
        raise NotImplementedError("pds_fusion_method %s" % method)
    flags = _ints(cfg["pds_fusion_layers"])
    stages = [i for i, f in enumerate(flags) if f]
    n = min(int(cfg["pds_stages"]), len(stages))
    return stages if n > 1 else []


def pds_encoder_forward(src_tokens, src_lengths, W, cfg, training=False, prefix="encoder.", bn_stats=None):
    """models/speech_to_text/pdss2t_transformer.py:1042-1281 (no fusion / inter-CTC / mixup):
    pad T up to the NEXT multiple of prod(ratios) (always pads: +prod when already aligned, :1050-1055);
    per stage: mask -> Conv1d(k, stride r, pad (k-1)//2) -> LayerNorm -> mask (Downsampling.forward :107-144),
    lengths floor((l-1)/r + 1); + sinusoidal positions (or rel-pos table); layers = PDSTransformerEncoderLayer
    (modules/pds_layer.py:263-359: the S2T layer, conv-module activation = encoder_activation_fn)."""
    ratios, dims = _ints(cfg["pds_ratios"]), _ints(cfg["pds_embed_dims"])
    layers, heads = _ints(cfg["pds_layers"]), _ints(cfg["pds_attn_heads"])
    ksz, ffr = _ints(cfg["pds_kernel_sizes"]), _ints(cfg["pds_ffn_ratios"])
    posf = _ints(cfg["pds_position_embed"])
    B, T, _ = src_tokens.shape
    total = 1
    for r in ratios:
        total *= max(1, r)
    pad_to = total - T % total
    x = src_tokens
    if total > 1 and pad_to > 0:
        x = torch.cat([x, x.new_zeros(B, pad_to, x.size(2))], dim=1)
    lens = src_lengths
    lcfg = dict(cfg)
    lcfg["activation_fn"] = cfg.get("encoder_activation_fn", "relu")  # conv-module activation rule of PDS layers
    prev_state, prev_mask = [], []
    for i in range(int(cfg["pds_stages"])):
        st = i + 1
        Tn = x.size(1)
        mask = lengths_to_padding_mask(lens, Tn)
        x = x.masked_fill(mask[:, :, None], 0.0)
        w, b = W[f"{prefix}downsampling{st}.conv.0.weight"], W[f"{prefix}downsampling{st}.conv.0.bias"]
        r = ratios[i]
        x = F.conv1d(x.transpose(1, 2), w, b, stride=r, padding=(ksz[i] - 1) // 2).transpose(1, 2)
        lens = torch.floor((lens.float() - 1) / r + 1).long()
        if cfg.get("pds_embed_norm", False):
            x = layer_norm(x, W[f"{prefix}downsampling{st}.norm.weight"], W[f"{prefix}downsampling{st}.norm.bias"])
        Tn = x.size(1)
        mask = lengths_to_padding_mask(lens, Tn)
        x = x.masked_fill(mask[:, :, None], 0.0)
        pos_tab = None
        if posf[i]:
            if cfg["encoder_attention_type"] == "rel_pos":
                pos_tab = rel_pos_table(Tn, dims[i])
            else:
                x = x + sinusoidal_positions(mask, dims[i], padding_idx=1)
        lcfg["encoder_attention_heads"] = heads[i]
        for j in range(layers[i]):
            x = encoder_layer(x, mask, pos_tab, W, f"{prefix}stage{st}.{j}.", lcfg, training, bn_stats)
        prev_state.append(x)
        prev_mask.append(mask)
    fusion = pds_fusion_stages(cfg)
    if fusion:
        # multi-scale representation fusion (:1187-1233, ``all_conv2``): every flagged stage output -> pre LayerNorm ->
        # DownSampleConvolutionModule (modules/downsample_convolution.py:75-123: mask, pointwise conv (bias), depthwise
        # conv kernel = stride = the remaining down-sampling ratio (bias, no padding), BatchNorm, Swish, pointwise conv
        # (bias), mask at floor(len / stride)) -> post LayerNorm; x = sum_i fusion_weight_i * state_i
        # fixed weights from --pds-fusion-weight, or the learned parameter fusion_weight (:799-809)
        fw_cfg = cfg.get("pds_fusion_weight", None)
        fw = [float(t) for t in str(fw_cfg).split("_")] if fw_cfg not in (None, "None") else W[prefix + "fusion_weight"]
        acc = None
        for n_f, i in enumerate(fusion):
            st = i + 1
            s_, m_ = prev_state[i], prev_mask[i]
            if cfg.get("pds_fusion_mask", False):
                s_ = s_.masked_fill(m_[:, :, None], 0.0)
            if not cfg.get("pds_fusion_no_prenorm", False):
                s_ = layer_norm(s_, W[f"{prefix}fusion_pre_layer_norm{st}.weight"], W[f"{prefix}fusion_pre_layer_norm{st}.bias"])
            q = f"{prefix}fusion_downsampling{st}."
            s_ = s_.masked_fill(m_[:, :, None], 0.0)
            y = linear(s_, W[q + "pointwise_conv1.weight"][:, :, 0], W[q + "pointwise_conv1.bias"])
            wd = W[q + "depthwise_conv.weight"]
            r = wd.size(2)
            y = F.conv1d(y.transpose(1, 2), wd, W[q + "depthwise_conv.bias"], stride=r, groups=wd.size(0)).transpose(1, 2)
            if training:
                mean = y.mean(dim=(0, 1))
                var = ((y - mean) ** 2).mean(dim=(0, 1))
                if bn_stats is not None:
                    bn_stats[q[:-1] + ".norm"] = (mean, var, y.size(0) * y.size(1))
            else:
                mean, var = W[q + "norm.running_mean"], W[q + "norm.running_var"]
            y = (y - mean) / torch.sqrt(var + 1e-5) * W[q + "norm.weight"] + W[q + "norm.bias"]
            y = activation("swish", y)
            y = linear(y, W[q + "pointwise_conv2.weight"][:, :, 0], W[q + "pointwise_conv2.bias"])
            ol = ((~m_).sum(-1) / r).long()
            y = y.masked_fill(lengths_to_padding_mask(ol, y.size(1))[:, :, None], 0.0)
            y = layer_norm(y, W[f"{prefix}fusion_post_layer_norm{st}.weight"], W[f"{prefix}fusion_post_layer_norm{st}.bias"])
            acc = fw[n_f] * y if acc is None else acc + fw[n_f] * y
        x = acc
    x = layer_norm(x, W[prefix + "layer_norm.weight"], W[prefix + "layer_norm.bias"])
    out = {"encoder_out": [x.transpose(0, 1)], "encoder_padding_mask": [mask], "ctc_logit": []}
    if prefix + "ctc.ctc_projection.weight" in W:
        logit = linear(x, W[prefix + "ctc.ctc_projection.weight"], W[prefix + "ctc.ctc_projection.bias"])
        out["ctc_logit"] = [logit.transpose(0, 1)]
    return out


# ----------------------------------------------------------------------------------------------
# SATE stacked acoustic + textual encoder, a17
# ----------------------------------------------------------------------------------------------
def _layer_list(spec, n_layers):
    if spec is None or str(spec) in ("", "None", "none"):
        return []
    return [int(t) + n_layers if int(t) <= 0 else int(t) for t in str(spec).split(",")]


def sate_encoder_forward(src_tokens, src_lengths, W, cfg, training=False, prefix="encoder.", bn_stats=None,
                         ctc_alignment_oracle=None, oracle_masks=None, drop_self_attn=None):
    """models/speech_to_text/s2t_sate.py:973-1075: acoustic encoder (a6) -> adapter `inter_league`
    (modules/speech_to_text/adapter.py:214-217,264-266,296-297: x + softmax(ctc_logit / tau) @ W_embed) ->
    TextualEncoder.forward (:641-827): [embed LN], scale, [+ sinusoidal positions unless text_no_pos_emb], N layers,
    final LayerNorm, with the NAST extras of SURVEY.md §8f row 2:
      * layers from ``cross_attn_start_layer`` on are TransformerS2EncoderLayer (modules/transformer_s2_layer.py:214-336,
        ``serial``): self-attention block, then a second pre-LN attention block (``s2_attn_norm`` -> ``s2_attn`` with
        keys/values = ``attn_norm`` of layer ``cross_attn_layer``'s output) and the FFN block; in training the
        self-attention block is skipped with probability ``cross_attn_league_drop_net_prob`` (``drop_self_attn``: list of
        bools, one per S2 layer, overrides the draw);
      * XCTC head on the output (``xctc``), intermediate XCTC after ``inter_xctc_layers`` (``xctc_norm{L}`` -> shared
        head) followed by prediction-aware encoding ``xctc_pae`` (inter_league, with the ground-truth mixing of
        :774-797 when ``xctc_pae_ground_truth_ratio`` > 0 and an alignment oracle is passed)."""
    ac = encoder_forward(src_tokens, src_lengths, W, cfg, training, prefix + "acoustic_encoder.", bn_stats,
                         ctc_alignment_oracle, oracle_masks)
    x = ac["encoder_out"][0].transpose(0, 1)  # (B, T', d)
    mask = ac["encoder_padding_mask"][0]
    d = x.size(-1)
    if cfg.get("adapter", "none") == "inter_league":
        x = pae_inter_league(x, ac["ctc_logit"][0].transpose(0, 1), W[prefix + "adapter.embed_adapter.weight"],
                             cfg.get("adapter_temperature", 1.0))
    elif cfg.get("adapter", "none") != "none":
        raise NotImplementedError(cfg["adapter"])
    p = prefix + "textual_encoder."
    if cfg.get("textual_encoder_embed_norm", False):
        x = layer_norm(x, W[p + "embed_ln.weight"], W[p + "embed_ln.bias"])
    if not cfg.get("textual_encoder_no_scale_embedding", False):
        x = x * math.sqrt(d)
    if not cfg.get("text_no_pos_emb", False):
        x = x + sinusoidal_positions(mask, d, padding_idx=1)
    h = cfg["encoder_attention_heads"]
    n_layers = int(cfg["text_encoder_layers"])
    use_xctc = float(cfg.get("xctc_weight", 0) or 0) > 0
    inter_x = _layer_list(cfg.get("inter_xctc_layers"), n_layers) if float(cfg.get("inter_xctc_weight", 0) or 0) > 0 else []
    cross = bool(cfg.get("xctc_cross_attn", False)) and cfg.get("cross_attn_start_layer") is not None
    start, src_layer = int(cfg.get("cross_attn_start_layer", 0) or 0), int(cfg.get("cross_attn_layer", 0) or 0)
    gt = float(cfg.get("xctc_pae_ground_truth_ratio", 0) or 0)
    xh = p + "xctc.ctc_projection."
    attn_x, xorc, x_force_emit, inter_xlogits, s2_i = None, None, None, [], 0
    for i in range(n_layers):
        q = f"{p}layers.{i}."
        s2 = attn_x if (cross and i >= start - 1) else None
        y = layer_norm(x, W[q + "self_attn_layer_norm.weight"], W[q + "self_attn_layer_norm.bias"])
        sa = mha(y, y, W, q + "self_attn.", h, mask)
        is_s2_layer = cross and i >= start - 1
        skip = False
        if is_s2_layer and training and cfg.get("cross_attn_league_drop_net", False):
            if drop_self_attn is not None:
                skip = bool(drop_self_attn[s2_i])
            else:
                skip = float(np.random.uniform(0, 1)) < float(cfg.get("cross_attn_league_drop_net_prob", 0))
        if is_s2_layer:
            s2_i += 1
        if not skip:
            x = x + sa
        if s2 is not None:  # serial collaboration (:281-296)
            y = layer_norm(x, W[q + "s2_attn_norm.weight"], W[q + "s2_attn_norm.bias"])
            x = x + mha(y, s2, W, q + "s2_attn.", h, mask)
        y = layer_norm(x, W[q + "final_layer_norm.weight"], W[q + "final_layer_norm.bias"])
        x = x + ffn(y, W, q, cfg.get("activation_fn", "relu"), n1="fc1", n2="fc2")
        L = i + 1
        if cross and L == src_layer:
            attn_x = layer_norm(x, W[p + "attn_norm.weight"], W[p + "attn_norm.bias"])
        if L in inter_x:
            npre = p + ("layer_norm." if cfg.get("share_inter_xctc_norm", False) else "xctc_norm%d." % L)
            nx = layer_norm(x, W[npre + "weight"], W[npre + "bias"])
            logit = linear(nx, W[xh + "weight"], W[xh + "bias"])
            orc = msk = None
            if gt > 0 and ctc_alignment_oracle is not None and ctc_alignment_oracle.get("xctc") is not None:
                if xorc is None:
                    xorc = pae_oracle_mask(ctc_alignment_oracle["xctc"], gt,
                                           cfg.get("xctc_pae_ground_truth_ratio_adaptive", False),
                                           cfg.get("xctc_pae_ground_truth_only_mistake", False),
                                           (oracle_masks or {}).get("xctc"))
                    x_force_emit = xorc[2]
                orc, msk = xorc[0], xorc[1]
            if cfg.get("xctc_pae", "none") == "inter_league":
                x = pae_inter_league(x if cfg.get("pae_unnorm_input", False) else nx, logit,
                                     W[p + "xctc_pae.embed_adapter.weight"], cfg.get("pae_ctc_temperature", 1.0), orc, msk,
                                     cfg.get("pae_oracle_smooth", False))
            elif cfg.get("xctc_pae", "none") != "none":
                raise NotImplementedError(cfg["xctc_pae"])
            inter_xlogits.append(logit.transpose(0, 1))
    x = layer_norm(x, W[p + "layer_norm.weight"], W[p + "layer_norm.bias"])
    out = {"encoder_out": [x.transpose(0, 1)], "encoder_padding_mask": [mask], "ctc_padding_mask": [mask],
           "ctc_logit": ac["ctc_logit"], "inter_ctc_logits": ac.get("inter_ctc_logits", []), "xctc_logit": [],
           "inter_xctc_logits": inter_xlogits, "ctc_force_emit": ac.get("ctc_force_emit"), "xctc_force_emit": x_force_emit}
    if use_xctc:
        out["xctc_logit"] = [linear(x, W[xh + "weight"], W[xh + "bias"]).transpose(0, 1)]
    return out


def ctc_align_oracle(logit_tbv, tokens, input_lengths, pad_idx=1, eos_idx=2, blank=0):
    """criterions/ctc.py:286-312 (get_ctc_align): Viterbi CTC alignment of the reference text (torch_imputer
    best_alignment), padded with state 0; oracle label per frame (blank on even states); mistake_flag = the model's
    arg-max differs from the oracle label; mistake_ratio = mistakes / input length.  tokens (B,U) with pad/eos."""
    lp = torch.log_softmax(logit_tbv.float(), dim=-1)
    T, B, _ = lp.shape
    keep = (tokens != pad_idx) & (tokens != eos_idx)
    tlens = keep.sum(-1)
    aligns = best_alignment(lp, [tokens[b][: int(tlens[b])].tolist() for b in range(B)], input_lengths, blank)
    pad = torch.tensor([list(a) + [0] * (T - len(a)) for a in aligns], dtype=tokens.dtype)
    pos = torch.div(pad, 2, rounding_mode="floor").clip(max=tokens.shape[1] - 1)
    oracle = tokens.gather(-1, pos)
    oracle = oracle.masked_fill(pad % 2 == 0, blank)
    mistake_flag = lp.argmax(-1).transpose(0, 1) != oracle
    mistake_ratio = mistake_flag.sum(-1) / input_lengths
    return oracle, pad, mistake_flag, mistake_ratio


def ctc_criterion_loss(W, cfg, src_tokens, src_lengths, target, transcript=None, training=True, weights=None,
                       bn_stats=None, oracle_masks=None, drop_self_attn=None, drop_self_attn_first=None, pad_idx=1,
                       eos_idx=2, blank=0):
    """criterions/ctc.py:258-281 (forward) + :542-1016 (compute_ctc_loss) for an encoder-only model (s2t_ctc):
    loss = ctc_weight * CTC(ctc_logit, transcript) + inter_ctc_weight * mean_i CTC(inter_ctc_logit_i, transcript)
         + xctc_weight * CTC(xctc_logit, target) + inter_xctc_weight * mean_i CTC(inter_xctc_logit_i, target)
    (summed over utterances, zero_infinity; transcript defaults to target).  In training with a PAE ground-truth ratio,
    a first no-grad pass produces the alignment oracle (:283-433).  ``weights``: dict with ctc/inter_ctc/xctc/inter_xctc.
    Returns (loss, log, encoder_out)."""
    w = {"ctc": float(cfg.get("ctc_weight", 0) or 0), "inter_ctc": float(cfg.get("inter_ctc_weight", 0) or 0),
         "xctc": float(cfg.get("xctc_weight", 0) or 0), "inter_xctc": float(cfg.get("inter_xctc_weight", 0) or 0)}
    w.update(weights or {})
    fwd = ENCODERS[encoder_kind(cfg)]
    transcript = target if transcript is None else transcript
    gt = float(cfg.get("ctc_pae_ground_truth_ratio", 0) or 0) + float(cfg.get("xctc_pae_ground_truth_ratio", 0) or 0)
    kw = {}
    if encoder_kind(cfg) == "sate":
        kw["drop_self_attn"] = drop_self_attn
    if training and gt != 0:
        with torch.no_grad():
            kw1 = dict(kw, drop_self_attn=drop_self_attn_first) if "drop_self_attn" in kw else kw
            first = fwd(src_tokens, src_lengths, W, cfg, training=training, bn_stats=None, **kw1)
        in_lens = (~first["encoder_padding_mask"][0]).sum(-1)
        orc = {}
        if len(first["ctc_logit"]) > 0:
            orc["ctc"] = ctc_align_oracle(first["ctc_logit"][0], transcript, in_lens, pad_idx, eos_idx, blank)
        if len(first.get("xctc_logit", [])) > 0:
            orc["xctc"] = ctc_align_oracle(first["xctc_logit"][0], target, in_lens, pad_idx, eos_idx, blank)
        kw.update(ctc_alignment_oracle=orc, oracle_masks=oracle_masks)
    enc = fwd(src_tokens, src_lengths, W, cfg, training=training, bn_stats=bn_stats, **kw)
    in_lens = (~enc["encoder_padding_mask"][0]).sum(-1)

    def one(logit_tbv, toks, lens=None):
        return ctc_nll(torch.log_softmax(logit_tbv.float(), -1), ctc_targets(toks, pad_idx, eos_idx),
                       in_lens if lens is None else lens, blank).sum()

    log, loss = {}, 0.0
    if w["inter_ctc"] > 0 and len(enc.get("inter_ctc_logits", [])) > 0:
        masks = enc.get("inter_ctc_padding_masks")
        log["inter_ctc_loss"] = sum(one(l, transcript, None if masks is None else (~masks[j]).sum(-1))
                                    for j, l in enumerate(enc["inter_ctc_logits"])) / len(enc["inter_ctc_logits"])
        loss = loss + w["inter_ctc"] * log["inter_ctc_loss"]
    if w["ctc"] > 0 and len(enc["ctc_logit"]) > 0:
        log["ctc_loss"] = one(enc["ctc_logit"][0], transcript)
        loss = loss + w["ctc"] * log["ctc_loss"]
    if w["inter_xctc"] > 0 and len(enc.get("inter_xctc_logits", [])) > 0:
        log["inter_xctc_loss"] = sum(one(l, target) for l in enc["inter_xctc_logits"]) / len(enc["inter_xctc_logits"])
        loss = loss + w["inter_xctc"] * log["inter_xctc_loss"]
    if w["xctc"] > 0 and len(enc.get("xctc_logit", [])) > 0:
        log["xctc_loss"] = one(enc["xctc_logit"][0], target)
        loss = loss + w["xctc"] * log["xctc_loss"]
    return loss, log, enc


ENCODERS = {"s2t_transformer": encoder_forward, "pds": pds_encoder_forward, "sate": sate_encoder_forward}


def encoder_kind(cfg):
    arch = str(cfg.get("arch", ""))
    etype = str(cfg.get("encoder_type", "")) if arch.startswith("s2t_ctc") else ""  # s2t_ctc.py:60-68
    if arch.startswith("pdss2t") or etype == "pds":
        return "pds"
    if arch.startswith("s2t_sate") or etype == "sate":
        return "sate"
    return "s2t_transformer"


# ----------------------------------------------------------------------------------------------
# imputer loss / best alignment (a21) — restated from fairseq/torch_imputer/{imputer,best_alignment}.cu
# ----------------------------------------------------------------------------------------------
def _ext_labels(y, blank):
    L = 2 * len(y) + 1
    ext = [blank] * L
    ext[1::2] = list(y)
    return ext


def imputer_nll(log_probs, targets, force_emits, input_lengths, blank=0):
    """torch_imputer/imputer.cu:57-215 — CTC alpha recursion where frame t, when force_emits[b][t] >= 0, may only
    occupy that extended-label state (every other state -inf).  Pure-Python loops (small cases only).
    log_probs (T,B,V); targets: list of label lists; force_emits (B,T).  Returns nll per utterance (inf kept)."""
    T, B, V = log_probs.shape
    out = []
    for b in range(B):
        ext = _ext_labels([int(v) for v in targets[b]], blank)
        L, Tb = len(ext), int(input_lengths[b])
        ninf = float("-inf")
        a = [ninf] * L
        for s in range(min(2, L)):
            a[s] = float(log_probs[0, b, ext[s]])
        fe = int(force_emits[b][0])
        if fe > -1:
            a = [v if s == fe else ninf for s, v in enumerate(a)]
        for t in range(1, Tb):
            fe = int(force_emits[b][t])
            n = [ninf] * L
            for s in range(L):
                if fe > -1 and fe != s:
                    continue
                c = [a[s]]
                if s > 0:
                    c.append(a[s - 1])
                if s > 1 and ext[s] != blank and ext[s] != ext[s - 2]:
                    c.append(a[s - 2])
                m = max(c)
                if m == ninf:
                    continue
                n[s] = m + math.log(sum(math.exp(v - m) for v in c)) + float(log_probs[t, b, ext[s]])
            a = n
        c = [a[L - 1]] + ([a[L - 2]] if L > 1 else [])
        m = max(c)
        out.append(float("inf") if m == ninf else -(m + math.log(sum(math.exp(v - m) for v in c))))
    return torch.tensor(out)


def best_alignment(log_probs, targets, input_lengths, blank=0):
    """torch_imputer/best_alignment.cu:57-201 + imputer.py:245-259: Viterbi over CTC states (first max in the order
    s, s-1, s-2), backtrace from argmax(alpha[T-1, L-2:]) (first max; state 0 when L == 1)."""
    T, B, V = log_probs.shape
    res = []
    for b in range(B):
        ext = _ext_labels([int(v) for v in targets[b]], blank)
        L, Tb = len(ext), int(input_lengths[b])
        ninf = float("-inf")
        a = [ninf] * L
        for s in range(min(2, L)):
            a[s] = float(log_probs[0, b, ext[s]])
        bp = [[s for s in range(L)]]
        for t in range(1, Tb):
            n, p = [ninf] * L, [0] * L
            for s in range(L):
                m, arg = a[s], s
                if s > 0 and a[s - 1] > m:
                    m, arg = a[s - 1], s - 1
                if s > 1 and ext[s] != blank and ext[s] != ext[s - 2] and a[s - 2] > m:
                    m, arg = a[s - 2], s - 2
                n[s] = m + float(log_probs[t, b, ext[s]])
                p[s] = arg
            a = n
            bp.append(p)
        cur = 0 if L == 1 else (L - 1 if a[L - 1] > a[L - 2] else L - 2)
        path = [cur]
        for t in range(Tb - 1, 0, -1):
            cur = bp[t][cur]
            path.append(cur)
        res.append(path[::-1])
    return res


# ----------------------------------------------------------------------------------------------
# helpers for tests / bench
# ----------------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------------
# Row a22: the update after backward (one rank) — scale, clip, Adam, inverse-sqrt schedule
# ----------------------------------------------------------------------------------------------
def inverse_sqrt_lr(num_updates, lr, warmup_updates, warmup_init_lr):
    """optim/lr_scheduler/inverse_square_root_schedule.py:59-85: the rate SET by step_update(num_updates), i.e. the one
    the NEXT update runs with (trainer.py:802 calls it after the optimizer step; _build_optimizer calls it with 0)."""
    if num_updates < warmup_updates:
        return warmup_init_lr + num_updates * (lr - warmup_init_lr) / warmup_updates
    return lr * warmup_updates ** 0.5 * num_updates ** -0.5


def clip_coef(grads, max_norm):
    """utils.clip_grad_norm_ (utils.py:328-369): total = || per-tensor fp32 norms ||_2, coef = min(1, max / (total + 1e-6))."""
    total = torch.norm(torch.stack([torch.norm(g.detach().float()) for g in grads]))
    return total, float((max_norm / (total + 1e-6)).clamp(max=1.0))


def adam_update(p, g, m, v, t, lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0):
    """optim/adam.py:146-226 (fairseq's Adam, AdamW-style decay): in-place update number t >= 1 of one tensor."""
    b1, b2 = betas
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = v.sqrt().add_(eps)
    step = lr * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
    if weight_decay != 0:
        p.add_(p, alpha=-weight_decay * lr)
    p.addcdiv_(m, denom, value=-step)


def train_trajectory(W, cfg, src_tokens, src_lengths, prev_output_tokens, target, sample_size, updates, lr, betas, eps,
                     weight_decay, clip_norm, warmup_updates, warmup_init_lr):
    """``updates`` updates in the order of Trainer.train_step (trainer.py:611-759) on one rank: backward of the joint loss,
    grads * (1 / sample_size) (:729-734), clip (:737), Adam with the rate step_update set after the previous update.
    BatchNorm running statistics follow nn.BatchNorm1d (momentum 0.1, unbiased variance).  Returns per-update loss, norm, lr."""
    params = {k: t for k, t in W.items() if t.is_floating_point() and t.requires_grad}
    # tied weights (share_decoder_input_output_embed: transformer.py:918-933, share_ctc_and_embed: s2t_transformer.py:965-971)
    # are ONE parameter in the reference: one summed gradient, one Adam state.  The state dict lists them under every name.
    alias = {}
    tie = [k for k in ("decoder.embed_tokens.weight", "decoder.output_projection.weight", "encoder.ctc.ctc_projection.weight")
           if k in params]
    for k in tie[1:]:
        if params[k].shape == params[tie[0]].shape and torch.equal(params[k], params[tie[0]]):
            alias[k] = tie[0]
    m = {k: torch.zeros_like(t) for k, t in params.items() if k not in alias}
    v = {k: torch.zeros_like(t) for k, t in params.items() if k not in alias}
    losses, gnorms, lrs = [], [], []
    for n in range(updates):
        for t in params.values():
            t.grad = None
        bn = {}
        loss, _ = joint_loss(W, cfg, src_tokens, src_lengths, prev_output_tokens, target, eps=0.1, training=True,
                             use_torch_ctc=True, bn_stats=bn)
        loss.backward()
        with torch.no_grad():
            grads = {}
            for k, t in params.items():
                if t.grad is not None:
                    root = alias.get(k, k)
                    grads[root] = grads.get(root, 0) + t.grad / float(sample_size)
            total, coef = clip_coef(list(grads.values()), clip_norm)
            cur = inverse_sqrt_lr(n, lr, warmup_updates, warmup_init_lr)
            for k, g in grads.items():
                adam_update(params[k], g * coef, m[k], v[k], n + 1, cur, betas, eps, weight_decay)
            for k, root in alias.items():
                params[k].copy_(params[root])
            for name, (mean, var, cnt) in bn.items():
                W[name + ".running_mean"].mul_(0.9).add_(0.1 * mean)
                W[name + ".running_var"].mul_(0.9).add_(0.1 * var * cnt / (cnt - 1))
        losses.append(float(loss))
        gnorms.append(float(total))
        lrs.append(cur)
    return losses, gnorms, lrs


def cfg_from_golden(z) -> dict:
    cfg = {}
    for k in z.files:
        if k.startswith("cfg::"):
            v = z[k]
            name = k[5:]
            if v.dtype.kind in "US":
                cfg[name] = str(v)
            elif v.dtype == bool:
                cfg[name] = bool(v)
            else:
                f = float(v)
                cfg[name] = int(f) if f == int(f) and "weight" not in name and "dropout" not in name and "temperature" not in name else f
    cfg.setdefault("encoder_embed_norm", True)
    cfg.setdefault("encoder_no_scale_embedding", True)
    return cfg


def weights_from_golden(z, requires_grad=False) -> Dict[str, torch.Tensor]:
    W = {}
    for k in z.files:
        if k.startswith("w::"):
            t = torch.from_numpy(z[k])
            if requires_grad and t.is_floating_point():
                t = t.clone().requires_grad_(True)
            W[k[3:]] = t
    return W
