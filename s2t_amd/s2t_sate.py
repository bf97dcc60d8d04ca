"""SATE (stacked acoustic-and-textual encoding) on the HIP path — BASELINE config 4.

Reference: fairseq/models/speech_to_text/s2t_sate.py (S2TSATEModel :37-330, TextualEncoder :333-835, S2TSATEEncoder
:837-1125, architectures :1128-1387), fairseq/modules/speech_to_text/adapter.py:17-349 (`inter_league`),
fairseq/modules/transformer_layer.py:24-237 (TransformerEncoderLayer).
Built: the recipe configuration of egs/mustc/st/conf/sate.yaml (`acoustic-encoder transformer`, `adapter inter_league`,
selfattn textual encoder with positions, no XCTC / cross-layer attention / history); other options raise.
"""
import math

import torch
import torch.nn as nn

from . import functional as Fn
from .modules import TABLES, LayerNorm, Linear, MultiheadAttention
from .registry import register_model, register_model_architecture
from .s2t_transformer import (AddPositions, Embedding, S2TTransformerEncoder, S2TTransformerModel,
                              TransformerDecoderScriptable, _d, _SinPosHolder, _unsupported,
                              base_architecture as _s2t_base)


class Adapter(nn.Module):
    """modules/speech_to_text/adapter.py — `inter_league`: x + softmax(ctc_logit / tau) @ embed_adapter.weight."""

    def __init__(self, dim, adapter_type, dictionary_size, embed_tokens=None, strategy=None):
        super().__init__()
        if adapter_type not in ("inter_league", "none"):
            raise NotImplementedError("adapter %s (HIP path: inter_league, none)" % adapter_type)
        self.adapter_type = adapter_type
        if adapter_type == "inter_league":
            self.embed_adapter = Embedding(dictionary_size, dim, padding_idx=1) if embed_tokens is None else embed_tokens
        self.temperature = float((strategy or {}).get("distribution_temperature", 1.0))

    def forward(self, x2d, logit2d):
        if self.adapter_type == "none":
            return x2d
        return Fn.adapter_inter_league(x2d, logit2d, self.embed_adapter.weight, self.temperature)


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
        y, x = self.self_attn_layer_norm(x, fork=True)
        x = self.self_attn(y, None, x, B, T, T, lens)
        y, x = self.final_layer_norm(x, fork=True)
        return Fn.ffn(y, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                      self.activation_fn, 1.0, x, self.activation_dropout_p, self.dropout_p, self.training)


class TextualEncoder(nn.Module):
    """s2t_sate.py:333-835 — embed LN, scale, + sinusoidal positions, N layers, final LN."""

    def __init__(self, args, dictionary, embed_tokens=None):
        super().__init__()
        _unsupported(args, text_no_pos_emb=False, text_use_s2t_layer=False, xctc_weight=0, inter_xctc_layers=None,
                     axctc_weight=0)
        if getattr(args, "text_attention_type", "selfattn") != "selfattn":
            raise NotImplementedError("text attention type")
        d = args.encoder_embed_dim
        self.embed_dim = d
        self.register_buffer("version", torch.tensor([3.0]))
        self.embed_tokens = embed_tokens if embed_tokens is not None else Embedding(len(dictionary), d, 1)
        self.embed_scale = 1.0 if getattr(args, "textual_encoder_no_scale_embedding", False) else math.sqrt(d)
        self.embed_ln = LayerNorm(d) if getattr(args, "textual_encoder_embed_norm", False) else None
        self.embed_positions = _SinPosHolder()
        self.layers = nn.ModuleList([TransformerEncoderLayer(args) for _ in range(args.text_encoder_layers)])
        self.layer_norm = LayerNorm(d) if args.encoder_normalize_before else None
        self.max_pos = getattr(args, "max_source_positions", 6000)
        self.dropout_p = float(args.dropout or 0.0)

    def forward(self, x, B, T, lens32):
        if self.embed_ln is not None:
            x = self.embed_ln(x)
        tab = TABLES.get("sin", max(self.max_pos, T) + 2, self.embed_dim, x.device)
        x = AddPositions.apply(x, tab, lens32, T, self.embed_scale)
        x = Fn.dropout(x, self.dropout_p, self.training)  # dropout_module (s2t_sate.py:650)
        for layer in self.layers:
            x = layer(x, B, T, lens32)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        return x


class S2TSATEEncoder(nn.Module):
    """s2t_sate.py:837-1125."""

    def __init__(self, args, task=None, decoder_embed_tokens=None):
        super().__init__()
        _unsupported(args, use_enc_dlcl=False, freeze_acoustic_encoder=False, freeze_textual_encoder=False)
        if getattr(args, "acoustic_encoder", "transformer") != "transformer":
            raise NotImplementedError("acoustic encoder %s" % args.acoustic_encoder)
        self.acoustic_encoder = S2TTransformerEncoder(args, task, decoder_embed_tokens)
        vocab = len(task.source_dictionary)
        strategy = {"distribution_temperature": getattr(args, "adapter_temperature", 1.0)}
        self.adapter = Adapter(args.encoder_embed_dim, getattr(args, "adapter", "none"), vocab, strategy=strategy)
        # the reference ties the text embedding to the decoder's (s2t_sate.py build_model; golden state_dict)
        self.textual_encoder = TextualEncoder(args, task.source_dictionary, decoder_embed_tokens)
        self.compute_dtype = torch.float32

    def max_positions(self):
        return self.acoustic_encoder.max_positions()

    def set_num_updates(self, n):
        pass

    def forward(self, src_tokens, src_lengths=None, **kwargs):
        ac = self.acoustic_encoder(src_tokens, src_lengths)
        x_tbc = ac["encoder_out"][0]
        Tn, B, d = x_tbc.shape
        x = x_tbc.transpose(0, 1).reshape(B * Tn, d)
        mask = ac["encoder_padding_mask"][0]
        lens32 = (~mask).sum(1).to(torch.int32)
        if self.adapter.adapter_type != "none":
            logit = ac["ctc_logit"][0].transpose(0, 1).reshape(B * Tn, -1)
            x = self.adapter(x, logit)
        x = self.textual_encoder(x, B, Tn, lens32)
        return {
            "encoder_out": [x.view(B, Tn, d).transpose(0, 1)],
            "ctc_logit": ac["ctc_logit"],
            "inter_ctc_logits": [], "xctc_logit": [], "inter_xctc_logits": [], "axctc_logit": [], "inter_axctc_logits": [],
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
    _s2t_base(args)


@register_model_architecture("s2t_sate", "s2t_sate_s")
def s2t_sate_s(args):
    _d(args, "encoder_embed_dim", 256)
    _d(args, "encoder_ffn_embed_dim", 256 * 8)
    _d(args, "encoder_attention_heads", 4)
    _d(args, "decoder_attention_heads", 4)
    base_architecture(args)
