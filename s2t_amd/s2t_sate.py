"""SATE (stacked acoustic-and-textual encoding) on the HIP path — BASELINE config 4.

Reference: fairseq/models/speech_to_text/s2t_sate.py (S2TSATEModel :37-330, TextualEncoder :333-835, S2TSATEEncoder
:837-1125, architectures :1128-1387), fairseq/modules/speech_to_text/adapter.py:17-349 (`inter_league`),
fairseq/modules/transformer_layer.py:24-237 (TransformerEncoderLayer).
Built: the recipe configurations of egs/mustc/st/conf/sate.yaml (`acoustic-encoder transformer`, `adapter inter_league`,
selfattn textual encoder) and of egs/mustc/st/conf/reproduction_nast.yaml (XCTC / intermediate XCTC
heads, prediction-aware encoding with the ground-truth curriculum, cross-layer attention with TransformerS2EncoderLayer,
modules/transformer_s2_layer.py:214-336); DLCL history, AXCTC and the remaining options raise.
"""
import copy
import math

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn
from . import rows as Rows
from .modules import TABLES, Adapter, CTC, LayerNorm, Linear, MultiheadAttention, pae_oracle_mask
from .registry import register_model, register_model_architecture
from .s2t_transformer import (AddPositions, Embedding, S2TTransformerEncoder, S2TTransformerModel,
                              TransformerDecoderScriptable, _d, _SinPosHolder, _unsupported,
                              base_architecture as _s2t_base)


class TransformerEncoderLayer(nn.Module):
    """modules/transformer_layer.py:24-237 (pre-LN): keys self_attn, self_attn_layer_norm, fc1, fc2, final_layer_norm."""

    def __init__(self, args):
        super().__init__()
        d = args.encoder_embed_dim
        if not args.encoder_normalize_before:
            raise NotImplementedError("post-LN textual encoder layers")
        self.dropout_p = float(args.dropout or 0.0)
        self.activation_dropout_p = float(getattr(args, "activation_dropout", 0) or 0.0)
        self.self_attn = MultiheadAttention(d, args.encoder_attention_heads, dropout=getattr(args, "attention_dropout", 0.0),
                                            self_attention=True)
        self.self_attn.out_dropout = self.dropout_p
        self.self_attn_layer_norm = LayerNorm(d)
        self.fc1 = Linear(d, args.encoder_ffn_embed_dim)
        self.fc2 = Linear(args.encoder_ffn_embed_dim, d)
        self.final_layer_norm = LayerNorm(d)
        self.activation_fn = getattr(args, "activation_fn", "relu")

    def forward(self, x, B, T, lens):
        # packed rows (s2t_amd/rows.py): ``lens`` then carries the geometry — key side and query side of the self-attention, the
        # live-row bound of the feed-forward block
        rows = lens if getattr(lens, "_pk", None) is not None else None
        x = self.self_attn(x, None, None, B, T, T, lens, norm=self.self_attn_layer_norm)
        # (the LayerNorm, both products, activation, dropouts and the residual in one launch where the row-block kernel
        # applies: d = 256 bf16, relu / swish, >= S2T_FFN_FUSED_MIN_ROWS rows; the LayerNorm + GEMM composition otherwise)
        return Fn.ffn_block(x, self.final_layer_norm.weight, self.final_layer_norm.bias, self.fc1.weight, self.fc1.bias,
                            self.fc2.weight, self.fc2.bias, self.activation_fn, 1.0, self.activation_dropout_p, self.dropout_p,
                            self.training, rows=rows)


class TransformerS2EncoderLayer(TransformerEncoderLayer):
    """modules/transformer_s2_layer.py:24-336, ``serial`` collaboration: the self-attention block, then a second pre-LN
    attention block whose keys/values are another layer's (normalised) output — ``s2_attn_norm`` -> ``s2_attn`` ->
    dropout -> residual — then the FFN block.  With ``league_drop_net`` the self-attention block is skipped in training
    with probability ``league_drop_net_prob`` (host-side ``numpy.random.uniform`` draw, as in the reference :277-279).
    ``s2_norm`` exists for checkpoint compatibility (the textual encoder never asks for it: ``s2_need_norm=False``)."""

    def __init__(self, args):
        super().__init__(args)
        d = args.encoder_embed_dim
        if getattr(args, "encoder_collaboration_mode", "serial") != "serial":
            raise NotImplementedError("encoder collaboration mode %s" % args.encoder_collaboration_mode)
        _unsupported(args, encoder_league_out_norm=False, encoder_league_gated=False, squeeze_excitation=False)
        if getattr(args, "encoder_use_s2_attn_norm", True):
            self.s2_norm = LayerNorm(d)
        self.s2_attn_norm = LayerNorm(d)
        self.s2_attn = MultiheadAttention(d, args.encoder_attention_heads, dropout=getattr(args, "attention_dropout", 0.0),
                                          self_attention=False)
        self.s2_attn.out_dropout = self.dropout_p
        self.league_drop_net = bool(getattr(args, "encoder_league_drop_net", False))
        self.league_drop_net_prob = float(getattr(args, "encoder_league_drop_net_prob", 0.0) or 0.0)

    def forward(self, x, B, T, lens, s2=None, skip_self_attn=None):
        skip = False
        if self.training and self.league_drop_net:
            draw = float(np.random.uniform(0, 1)) < self.league_drop_net_prob  # always drawn, like the reference
            skip = draw if skip_self_attn is None else bool(skip_self_attn)
        rows = lens if getattr(lens, "_pk", None) is not None else None  # packed rows: queries, keys and the other layer's output
        if not skip:
            x = self.self_attn(x, None, None, B, T, T, lens, norm=self.self_attn_layer_norm)
        if s2 is not None:
            y, x = self.s2_attn_norm(x, fork=True, rows=rows)
            x = self.s2_attn(y, s2, x, B, T, T, lens, q_rows=rows)
        # (the LayerNorm, both products, activation, dropouts and the residual in one launch where the row-block kernel
        # applies: d = 256 bf16, relu / swish, >= S2T_FFN_FUSED_MIN_ROWS rows; the LayerNorm + GEMM composition otherwise)
        return Fn.ffn_block(x, self.final_layer_norm.weight, self.final_layer_norm.bias, self.fc1.weight, self.fc1.bias,
                            self.fc2.weight, self.fc2.bias, self.activation_fn, 1.0, self.activation_dropout_p, self.dropout_p,
                            self.training, rows=rows)


def _layer_list(spec, n_layers):
    if spec is None or str(spec) in ("", "None", "none"):
        return []
    out = []
    for t in str(spec).split(","):
        L = int(t)
        assert L <= n_layers, (L, n_layers)
        out.append(L + n_layers if L <= 0 else L)
    return out


class TextualEncoder(nn.Module):
    """s2t_sate.py:333-835 — [embed LN], scale, [+ sinusoidal positions], N layers, final LN, plus the NAST extras
    (SURVEY.md §8f row 2): XCTC head on the output, intermediate XCTC heads (``xctc_norm{L}`` -> shared ``xctc``) with
    prediction-aware encoding ``xctc_pae`` and its ground-truth curriculum, cross-layer attention (layers from
    ``cross_attn_start_layer`` on are TransformerS2EncoderLayer attending to ``attn_norm`` of layer
    ``cross_attn_layer``'s output)."""

    def __init__(self, args, dictionary, embed_tokens=None, target_dictionary=None):
        super().__init__()
        _unsupported(args, text_use_s2t_layer=False, axctc_weight=0, inter_axctc_layers=None, inter_ctc_drop_prob=0,
                     xctc_pae_ground_truth_ratio_decay=None, share_pae_and_xctc=False, cross_attn_ctc_logit=False,
                     xctc_layer=0)
        if getattr(args, "text_attention_type", "selfattn") != "selfattn":
            raise NotImplementedError("text attention type")
        d = args.encoder_embed_dim
        n = args.text_encoder_layers
        self.embed_dim = d
        self.register_buffer("version", torch.tensor([3.0]))
        self.embed_tokens = embed_tokens if embed_tokens is not None else Embedding(len(dictionary), d, 1)
        self.embed_scale = 1.0 if getattr(args, "textual_encoder_no_scale_embedding", False) else math.sqrt(d)
        self.embed_ln = LayerNorm(d) if getattr(args, "textual_encoder_embed_norm", False) else None
        self.text_no_pos_emb = bool(getattr(args, "text_no_pos_emb", False))
        if not self.text_no_pos_emb:
            self.embed_positions = _SinPosHolder()
        self.layer_norm = LayerNorm(d) if (args.encoder_normalize_before and n > 0) else None
        self.max_pos = getattr(args, "max_source_positions", 6000)
        self.dropout_p = float(args.dropout or 0.0)
        tgt = target_dictionary if target_dictionary is not None else dictionary
        vocab = self.embed_tokens.weight.shape[0] if embed_tokens is not None else len(tgt)

        def xctc_head():
            head = CTC(d, dictionary_size=vocab, dropout=args.dropout)
            if embed_tokens is not None and getattr(args, "share_xctc_and_embed", False) \
                    and head.ctc_projection.weight.shape == embed_tokens.weight.shape:
                head.ctc_projection.weight = embed_tokens.weight
            return head

        # XCTC (s2t_sate.py:387-415)
        self.use_xctc = float(getattr(args, "xctc_weight", 0) or 0) > 0
        if self.use_xctc:
            self.xctc = xctc_head()
        # intermediate XCTC + PAE (:417-476)
        self.gt_ratio = float(getattr(args, "xctc_pae_ground_truth_ratio", 0) or 0)
        self.pae_unnorm_input = bool(getattr(args, "pae_unnorm_input", False))
        self.pae_adaptive_gt = bool(getattr(args, "xctc_pae_ground_truth_ratio_adaptive", False))
        self.pae_gt_only_mistake = bool(getattr(args, "xctc_pae_ground_truth_only_mistake", False))
        self.inter_xctc_layers = []
        if float(getattr(args, "inter_xctc_weight", 0) or 0) > 0:
            self.inter_xctc_layers = _layer_list(getattr(args, "inter_xctc_layers", None), n)
        if self.inter_xctc_layers:
            self.share_inter_xctc_norm = bool(getattr(args, "share_inter_xctc_norm", False))
            if not self.share_inter_xctc_norm:
                for L in self.inter_xctc_layers:
                    setattr(self, "xctc_norm%d" % L, LayerNorm(d))
            if not hasattr(self, "xctc"):
                self.xctc = xctc_head()
            strategy = {"embed_norm": getattr(args, "pae_embed_norm", False), "out_norm": getattr(args, "pae_out_norm", False),
                        "ctc_temperature": getattr(args, "pae_ctc_temperature", 1.0), "gumbel": getattr(args, "pae_gumbel", False),
                        "distribution_hard": getattr(args, "pae_distribution_hard", None), "gt_ratio": self.gt_ratio,
                        "oracle_smooth": getattr(args, "pae_oracle_smooth", False)}
            self.xctc_pae = Adapter(d, getattr(args, "xctc_pae", "none"), len(tgt), strategy=strategy)
        # cross-layer attention (:536-611)
        self.use_cross_attn = False
        layers = [TransformerEncoderLayer(args) for _ in range(n)]
        if getattr(args, "xctc_cross_attn", False) and getattr(args, "cross_attn_start_layer", None) is not None \
                and getattr(args, "cross_attn_layer", None) is not None:
            self.use_cross_attn = True
            self.cross_attn_start_layer = int(args.cross_attn_start_layer)
            self.cross_attn_layer = int(args.cross_attn_layer)
            self.attn_norm = LayerNorm(d)
            s2_args = copy.copy(args)
            for k in ("collaboration_mode", "league_s1_ratio", "league_s2_ratio", "league_drop_net", "league_drop_net_prob",
                      "league_drop_net_mix", "league_out_norm", "league_gated"):
                setattr(s2_args, "encoder_" + k, getattr(args, "cross_attn_" + k, None))
            _unsupported(s2_args, encoder_league_drop_net_mix=False)
            layers = layers[: self.cross_attn_start_layer - 1] + [
                TransformerS2EncoderLayer(s2_args) for _ in range(n - self.cross_attn_start_layer + 1)]
        self.layers = nn.ModuleList(layers)

    def packable(self):
        """Packed rows (s2t_amd/rows.py) through the textual layers: heads of 64; the NAST extras (XCTC heads, prediction-aware
        encoding, cross-layer attention) run on the row map as well unless the ground-truth curriculum is on (training with
        ``xctc_pae_ground_truth_ratio`` > 0 mixes (B, T) oracle labels in) or a decoder reads an intermediate head."""
        if len(self.layers) == 0 or self.layers[0].self_attn.num_heads * 64 != self.embed_dim:
            return False
        if self.inter_xctc_layers and ((self.gt_ratio > 0 and self.training) or getattr(self, "decode_inter_logits", False)):
            return False
        return True

    def forward(self, x, B, T, lens32, encoder_padding_mask=None, **kwargs):
        """x [B*T, d] -> (x, xctc_logit, inter_xctc_logits); logits are (T, B, V) views, inter entries follow the
        reference: plain tensor, or [logit, None, force_emit] under the ground-truth curriculum (:774-797).
        ``lens32`` with a packed geometry (``packable()`` configurations only): x holds packed rows and stays packed."""
        rows = lens32 if getattr(lens32, "_pk", None) is not None else None
        assert rows is None or self.packable()
        if self.embed_ln is not None:
            x = self.embed_ln(x, rows=rows)
        if not self.text_no_pos_emb:
            tab = TABLES.get("sin", max(self.max_pos, T) + 2, self.embed_dim, x.device)
            x = AddPositions.apply(x, tab, lens32, T, self.embed_scale)
            x = Fn.dropout(x, self.dropout_p, self.training)  # dropout_module (s2t_sate.py:650)
        elif self.embed_scale != 1.0:
            x = x * self.embed_scale
        skips = kwargs.get("drop_self_attn")  # test hook: replay of the reference's league drop-net draws
        packed_x = {"xctc": None, "inter": []}
        attn_x = xorc = x_force_emit = None
        inter_xctc_logits, s2_i = [], 0
        for i, layer in enumerate(self.layers):
            if self.use_cross_attn and i >= self.cross_attn_start_layer - 1:
                x = layer(x, B, T, lens32, s2=attn_x, skip_self_attn=None if skips is None else skips[s2_i])
                s2_i += 1
            else:
                x = layer(x, B, T, lens32)
            L = i + 1
            if self.use_cross_attn and L == self.cross_attn_layer:
                attn_x = self.attn_norm(x, rows=rows)
            if L in self.inter_xctc_layers:
                norm = self.layer_norm if self.share_inter_xctc_norm else getattr(self, "xctc_norm%d" % L)
                norm_x = norm(x, rows=rows)
                # (an intermediate head that feeds PAE emits compute-dtype logits, as in training — unless a decoder reads the
                # intermediate heads: CTCDecoder(--ctc-inter-logit) sets ``decode_inter_logits`` and they follow ctc_out_dtype)
                logit2d = self.xctc(norm_x, out_dtype=self.ctc_out_dtype if (self.xctc_pae.adapter_type == "none" or getattr(
                    self, "decode_inter_logits", False)) else None, rows=rows)
                if rows is not None:  # the (T, B, V) view only if somebody reads it; the packed rows ride in ``packed_x``
                    il = Rows.LazyList([lambda l2=logit2d: Rows.unpack(l2.contiguous(), rows).view(B, T, -1).transpose(0, 1)])
                    packed_x["inter"].append(logit2d)
                else:
                    il = logit2d.view(B, T, -1).transpose(0, 1)
                inter_logit = il
                orc = msk = None
                if self.gt_ratio > 0:
                    oracle = (kwargs.get("ctc_alignment_oracle") or {}).get("xctc")
                    if oracle is not None:
                        if xorc is None:
                            xorc = pae_oracle_mask(oracle, self.gt_ratio, self.pae_adaptive_gt, self.pae_gt_only_mistake,
                                                   (kwargs.get("pae_oracle_masks") or {}).get("xctc"))
                            x_force_emit = xorc[2]
                        orc, msk = xorc[0], xorc[1]
                        inter_logit = [il, None, x_force_emit]
                if self.xctc_pae.adapter_type != "none":
                    x = self.xctc_pae(x if self.pae_unnorm_input else norm_x, logit2d, orc, msk, rows=rows)
                inter_xctc_logits.append(inter_logit)
        if self.layer_norm is not None:
            x = self.layer_norm(x, rows=rows)
        xctc_logit = None
        if self.use_xctc and rows is not None:
            x2d = self.xctc(x, out_dtype=self.ctc_out_dtype, rows=rows)
            packed_x["xctc"] = x2d
            xctc_logit = Rows.LazyList([lambda: Rows.unpack(x2d.contiguous(), rows).view(B, T, -1).transpose(0, 1)])
        elif self.use_xctc:
            xctc_logit = self.xctc(x, out_dtype=self.ctc_out_dtype).view(B, T, -1).transpose(0, 1)
            if x_force_emit is not None:
                xctc_logit = [xctc_logit, None, x_force_emit]
        self.last_packed = packed_x if rows is not None else None  # (read by S2TSATEEncoder right after this call)
        return x, xctc_logit, inter_xctc_logits

    ctc_out_dtype = None  # None -> compute dtype; eval decoding sets fp32


class S2TSATEEncoder(nn.Module):
    """s2t_sate.py:837-1125."""

    def __init__(self, args, task=None, decoder_embed_tokens=None):
        super().__init__()
        _unsupported(args, use_enc_dlcl=False, freeze_acoustic_encoder=False, freeze_textual_encoder=False,
                     adapter_ground_truth_ratio=0, share_adapter_and_ctc=False, share_adapter_and_embed=False)
        if getattr(args, "acoustic_encoder", "transformer") != "transformer":
            raise NotImplementedError("acoustic encoder %s" % args.acoustic_encoder)
        if args.text_encoder_layers > 0:
            setattr(args, "disable_xctc", True)  # :841-842
        self.acoustic_encoder = S2TTransformerEncoder(args, task, decoder_embed_tokens)
        vocab = len(task.source_dictionary)
        strategy = {"ctc_temperature": getattr(args, "adapter_temperature", 1.0)}
        self.adapter = Adapter(args.encoder_embed_dim, getattr(args, "adapter", "none"), vocab, strategy=strategy)
        # the reference hands the decoder's embedding to the textual encoder (s2t_sate.py:906)
        self.textual_encoder = TextualEncoder(args, task.source_dictionary, decoder_embed_tokens,
                                              getattr(task, "target_dictionary", None))
        self.pae_ground_truth_ratio = (float(getattr(args, "ctc_pae_ground_truth_ratio", 0) or 0)
                                       + float(getattr(args, "adapter_ground_truth_ratio", 0) or 0)
                                       + float(getattr(args, "xctc_pae_ground_truth_ratio", 0) or 0))  # :857-861
        self.compute_dtype = torch.float32

    def max_positions(self):
        return self.acoustic_encoder.max_positions()

    def set_num_updates(self, n):
        self.acoustic_encoder.set_num_updates(n)

    def set_ctc_infer(self, ctc_infer, post_process, src_dict=None, tgt_dict=None, path=None):
        if hasattr(self.acoustic_encoder, "ctc"):
            self.acoustic_encoder.ctc.set_infer(ctc_infer, post_process, src_dict, path)
        if hasattr(self.textual_encoder, "xctc"):
            self.textual_encoder.xctc.set_infer(ctc_infer, post_process, tgt_dict, path)

    def forward(self, src_tokens, src_lengths=None, **kwargs):
        ac = self.acoustic_encoder(src_tokens, src_lengths, **kwargs)
        pk = ac.get("packed")
        if (pk is not None and self.textual_encoder.packable() and Rows.ENABLED
                and (self.adapter.adapter_type == "none" or pk.get("ctc_logit") is not None)):
            # Packed rows (s2t_amd/rows.py) straight through: the acoustic encoder's rows, the adapter and the textual layers
            # run on the frames only (s2t_sate.py:973-1075 computes them on the padded frames too; nothing downstream reads those:
            # the decoder masks them as keys, the losses stop at the lengths)
            rows_, B, Tn = pk["rows"], pk["B"], pk["T"]
            x = pk["encoder_out"]
            d = x.shape[1]
            mask = ac["encoder_padding_mask"][0]
            if self.adapter.adapter_type != "none":
                x = self.adapter(x, pk["ctc_logit"], rows=rows_)
            xdt = getattr(self, "xctc_out_dtype", None)
            self.textual_encoder.ctc_out_dtype = xdt if xdt is not None else self.acoustic_encoder.ctc_out_dtype
            x, xctc_logit, inter_xctc_logits = self.textual_encoder(x, B, Tn, rows_, mask, **kwargs)
            px = self.textual_encoder.last_packed or {"xctc": None, "inter": []}
            return {
                "encoder_out": Rows.LazyList([lambda: Rows.unpack(x, rows_).view(B, Tn, d).transpose(0, 1)]),
                "ctc_logit": ac["ctc_logit"], "inter_ctc_logits": ac.get("inter_ctc_logits", []),
                "xctc_logit": [] if xctc_logit is None else xctc_logit, "inter_xctc_logits": inter_xctc_logits,
                "axctc_logit": [], "inter_axctc_logits": [], "ctc_padding_mask": [mask], "encoder_padding_mask": [mask],
                "mixup": None, "encoder_embedding": [], "encoder_states": [], "src_tokens": [], "src_lengths": [],
                "packed": {"rows": rows_, "B": B, "T": Tn, "encoder_out": x, "ctc_logit": pk.get("ctc_logit"),
                           "inter_ctc_logit": pk.get("inter_ctc_logit") or [], "xctc_logit": px["xctc"],
                           "inter_xctc_logit": px["inter"]},
            }
        x_tbc = ac["encoder_out"][0]
        Tn, B, d = x_tbc.shape
        x = x_tbc.transpose(0, 1).reshape(B * Tn, d)
        mask = ac["encoder_padding_mask"][0]
        lens32 = (~mask).sum(1).to(torch.int32)
        if self.adapter.adapter_type != "none":
            ctc_logit = ac["ctc_logit"][0]
            logit = ctc_logit[0] if isinstance(ctc_logit, (list, tuple)) else ctc_logit  # :1008-1012
            x = self.adapter(x, logit.transpose(0, 1).reshape(B * Tn, -1))
        # dtype of the XCTC logits (the decoded head of the NAST stack): `xctc_out_dtype` when set, else what the acoustic
        # encoder's CTC head uses — a decoder that reads `xctc_logit` only can leave the acoustic CTC logits (which then feed
        # nothing but the adapter's softmax) in the compute dtype
        xdt = getattr(self, "xctc_out_dtype", None)
        self.textual_encoder.ctc_out_dtype = xdt if xdt is not None else self.acoustic_encoder.ctc_out_dtype
        x, xctc_logit, inter_xctc_logits = self.textual_encoder(x, B, Tn, lens32, mask, **kwargs)
        return {
            "encoder_out": [x.view(B, Tn, d).transpose(0, 1)],
            "ctc_logit": ac["ctc_logit"],
            "inter_ctc_logits": ac.get("inter_ctc_logits", []),
            "xctc_logit": [] if xctc_logit is None else [xctc_logit],
            "inter_xctc_logits": inter_xctc_logits, "axctc_logit": [], "inter_axctc_logits": [],
            "ctc_padding_mask": [mask],
            "encoder_padding_mask": [mask],
            "mixup": None, "encoder_embedding": [], "encoder_states": [], "src_tokens": [], "src_lengths": [],
        }

    def reorder_encoder_out(self, encoder_out, new_order):
        out = S2TTransformerEncoder.reorder_encoder_out(self, encoder_out, new_order)
        out["ctc_padding_mask"] = [m.index_select(0, new_order) for m in encoder_out.get("ctc_padding_mask", [])]
        return out


@register_model("s2t_sate")
class S2TSATEModel(S2TTransformerModel):
    """models/speech_to_text/s2t_sate.py:37-330."""

    _REF_NAME = "s2t_sate"

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)
        tgt = task.target_dictionary
        embed = Embedding(len(tgt), args.decoder_embed_dim, tgt.pad())
        return cls(S2TSATEEncoder(args, task, embed), TransformerDecoderScriptable(args, tgt, embed))


@register_model_architecture("s2t_sate", "s2t_sate")
def base_architecture(args):
    _d(args, "text_encoder_layers", 6)
    _d(args, "acoustic_encoder", "transformer")
    _d(args, "adapter", "league")
    _d(args, "adapter_temperature", 1.0)
    _d(args, "textual_encoder_embed_norm", False)
    _d(args, "textual_encoder_no_scale_embedding", False)
    _d(args, "text_attention_type", "selfattn")
    _d(args, "cross_attn_collaboration_mode", "serial")
    _d(args, "cross_attn_league_drop_net", False)
    _d(args, "cross_attn_league_drop_net_prob", 0.0)
    _s2t_base(args)


@register_model_architecture("s2t_sate", "s2t_sate_s")
def s2t_sate_s(args):
    _d(args, "encoder_embed_dim", 256)
    _d(args, "encoder_ffn_embed_dim", 256 * 8)
    _d(args, "encoder_attention_heads", 4)
    _d(args, "decoder_attention_heads", 4)
    base_architecture(args)
