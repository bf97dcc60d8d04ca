"""PDS (progressive down-sampling) S2T encoder on the HIP path — BASELINE config 3.

Reference: fairseq/models/speech_to_text/pdss2t_transformer.py (Downsampling :53-144, PDSS2TTransformerEncoder
ctor :290-960 / forward :1042-1281, architectures :1358-1723) and fairseq/modules/pds_layer.py:263-359.
Built: the recipe configuration of egs/mustc/asr/conf/pds_base_8.yaml (`pds-ds-method conv`, `pds-embed-norm`, sinusoidal or
rel-pos position embedding per stage, no fusion, no inter-CTC); other options raise NotImplementedError.
state_dict keys: downsampling{i}.conv.0.{weight,bias}, downsampling{i}.norm.*, pos_embed{i}._float_tensor, stage{i}.N.*,
layer_norm.*, ctc.ctc_projection.*.
"""
from functools import reduce

import torch
import torch.nn as nn

from . import functional as Fn
from . import rows as Rows
from .modules import (CTC, TABLES, Ctx, DownSampleConvolutionModule, LayerNorm, MaskRows, S2TTransformerEncoderLayer,
                      _Conv1dK)
from .registry import register_model, register_model_architecture
from .s2t_transformer import (AddPositions, Embedding, S2TTransformerModel, TransformerDecoderScriptable,
                              _SinPosHolder, _d, _unsupported, base_architecture as _s2t_base)


def _ints(v):
    return [int(t) for t in str(v).split("_")]


class Downsampling(nn.Module):
    """pdss2t_transformer.py:53-144 (`conv` way): mask -> Conv1d(k, stride, (k-1)//2) -> LayerNorm -> mask."""

    def __init__(self, reduced_way, embed_norm, in_channels, out_channels, kernel_sizes, stride, padding):
        super().__init__()
        if reduced_way != "conv" or padding != (kernel_sizes - 1) // 2 or stride < 1:
            raise NotImplementedError("PDS down-sampling: only the `conv` way of the recipes is built")
        self.stride = stride
        self.conv = nn.ModuleList([_Conv1dK(in_channels, out_channels, kernel_sizes)])
        self.norm = LayerNorm(out_channels) if embed_norm else None

    def forward(self, x2d, B, T, lens32):
        x2d = MaskRows.apply(x2d, lens32, T)
        c = self.conv[0]
        y = Fn.conv1d(x2d, c.weight, c.bias, B, T, self.stride)
        kk = c.weight.shape[1]
        Tout = (T + 2 * ((kk - 1) // 2) - kk) // self.stride + 1
        out_lens = torch.floor((lens32.float() - 1) / self.stride + 1).to(torch.int32)
        if self.norm is not None:
            y = self.norm(y, out_lens, Tout)  # LayerNorm + the trailing mask, fused
        else:
            y = MaskRows.apply(y, out_lens, Tout)
        return y, Tout, out_lens


class PDSS2TTransformerEncoder(nn.Module):
    def __init__(self, args, task=None, embed_tokens=None):
        super().__init__()
        _unsupported(args, inter_mixup=False, inter_ctc_layers=None, inter_xctc_layers=None, xctc_weight=0,
                     pds_final_layers=0)
        self.dropout_p = float(args.dropout or 0.0)
        self.pds_dropout_p = float(getattr(args, "pds_dropout", args.dropout) or 0.0)
        self.args = args
        self.pds_stages = int(args.pds_stages)
        self.pds_layers = _ints(args.pds_layers)
        self.pds_ratios = _ints(args.pds_ratios)
        self.pds_embed_dims = _ints(args.pds_embed_dims)
        self.pds_kernel_sizes = _ints(args.pds_kernel_sizes)
        self.pds_ffn_ratios = _ints(args.pds_ffn_ratios)
        self.pds_attn_heads = _ints(args.pds_attn_heads)
        self.pds_position_embed = _ints(args.pds_position_embed)
        self.attn_type = getattr(args, "encoder_attention_type", "selfattn")
        self.embed_dim = self.pds_embed_dims[-1]
        self.padding_idx = 1
        in_dim = args.input_feat_per_channel * args.input_channels
        for i in range(self.pds_stages):
            d = self.pds_embed_dims[i]
            ds = Downsampling(args.pds_ds_method, bool(args.pds_embed_norm), in_dim if i == 0 else self.pds_embed_dims[i - 1], d,
                              self.pds_kernel_sizes[i], self.pds_ratios[i], (self.pds_kernel_sizes[i] - 1) // 2)
            setattr(self, f"downsampling{i + 1}", ds)
            if self.pds_position_embed[i] and self.attn_type != "rel_pos":
                setattr(self, f"pos_embed{i + 1}", _SinPosHolder())
            # PDS layers take their conv-module activation from --encoder-activation-fn (pds_layer.py:67,104)
            stage = nn.ModuleList([
                S2TTransformerEncoderLayer(args, embed_dim=d, ffn_dim=d * self.pds_ffn_ratios[i], num_heads=self.pds_attn_heads[i],
                                           conv_activation=getattr(args, "encoder_activation_fn", "relu"))
                for _ in range(self.pds_layers[i])])
            setattr(self, f"stage{i + 1}", stage)
        # multi-scale representation fusion (pdss2t_transformer.py:357-391,588-640,1187-1233; ``all_conv2``): the flagged
        # stage outputs -> fusion_pre_layer_norm{i} -> DownSampleConvolutionModule down to the last stage's frame rate ->
        # fusion_post_layer_norm{i}; x = sum_i fusion_weight_i * state_i (fixed weights from --pds-fusion-weight, or the learned parameter)
        self.fusion_stages = []
        method = str(getattr(args, "pds_fusion_method", "none") or "none")
        if getattr(args, "pds_fusion", False) and method not in ("none", "None"):
            kind, _, transform = method.partition("_")
            if kind != "all" or (transform or "conv") != "conv2":
                raise NotImplementedError("--pds-fusion-method %s (HIP path: all_conv2)" % method)
            flags = _ints(args.pds_fusion_layers)
            stages = [i for i, f in enumerate(flags) if f]
            if min(self.pds_stages, len(stages)) > 1:
                self.fusion_stages = stages
                if getattr(args, "pds_fusion_weight", None) is None:  # learned (:799-806): a parameter, uniform at start
                    self.fusion_weight = torch.nn.Parameter(torch.full((len(stages),), 1.0 / len(stages)))
                else:
                    self.fusion_weight = [float(t) for t in str(args.pds_fusion_weight).split("_")]
                assert len(self.fusion_weight) == len(stages)
                self.fusion_mask = bool(getattr(args, "pds_fusion_mask", False))
                self.fusion_no_prenorm = bool(getattr(args, "pds_fusion_no_prenorm", False))
                for i in stages:
                    ratio = reduce(lambda a, b: a * b, self.pds_ratios[i + 1:], 1)
                    setattr(self, f"fusion_downsampling{i + 1}",
                            DownSampleConvolutionModule(self.embed_dim, kernel_size=ratio, stride=ratio,
                                                        input_channels=self.pds_embed_dims[i]))
                    setattr(self, f"fusion_pre_layer_norm{i + 1}", LayerNorm(self.pds_embed_dims[i]))
                    setattr(self, f"fusion_post_layer_norm{i + 1}", LayerNorm(self.embed_dim))
        self.layer_norm = LayerNorm(self.embed_dim) if args.encoder_normalize_before else None
        self.use_ctc = getattr(args, "ctc_weight", 0) > 0
        if self.use_ctc:
            if getattr(args, "ctc_layer", 0) != 0:
                raise NotImplementedError("ctc_layer inside the stack")
            vocab = len(task.source_dictionary) if task is not None else args.vocab_size
            self.ctc = CTC(self.embed_dim, dictionary_size=vocab, dropout=args.dropout)
            if getattr(args, "share_ctc_and_embed", False) and embed_tokens is not None \
                    and self.embed_dim == embed_tokens.embedding_dim:
                self.ctc.ctc_projection.weight = embed_tokens.weight
        self.compute_dtype = torch.float32
        self.ctc_out_dtype = None
        # rows kept behind every utterance of a packed stage: the reach of the conv module's depthwise kernel (s2t_amd/rows.py)
        self._halo = (int(getattr(args, "cnn_module_kernel", 31)) - 1) // 2 if getattr(args, "use_cnn_module", False) else 0

    def max_positions(self):
        return getattr(self.args, "max_source_positions", 6000)

    def set_num_updates(self, n):
        pass

    def forward(self, src_tokens, src_lengths, **kwargs):
        if not src_tokens.is_cuda:
            raise RuntimeError("s2t_amd runs on the GPU only; there is no CPU fallback")
        dt = self.compute_dtype
        B, T, C = src_tokens.shape
        total = reduce(lambda a, b: max(1, a) * max(1, b), self.pds_ratios)
        pad_to = total - T % total  # always pads, + total when already aligned (pdss2t_transformer.py:1050-1055)
        Tn = T + (pad_to if total > 1 else 0)
        x = torch.zeros(B, Tn, C, dtype=dt, device=src_tokens.device)
        x[:, :T].copy_(src_tokens)
        x = x.view(B * Tn, C)
        lens32 = src_lengths.to(torch.int32)
        states = []
        # the frame counts of every stage depend on the batch's lengths only: computed once per batch object (outside a captured
        # step), so that the packed geometry of a stage hangs on a tensor that stays (s2t_amd/rows.py)
        def stage_lens(sl):
            out, l = [], sl.to(torch.int32)
            for r in self.pds_ratios:
                l = torch.floor((l.float() - 1) / r + 1).to(torch.int32)
                out.append(l)
            return tuple(out)

        lens_memo = Fn.batch_memo(("pds_stage_lens", Fn.memo_owner(self)), (src_lengths,), stage_lens)
        packed_last = None
        for i in range(self.pds_stages):
            x, Tn, lens32 = getattr(self, f"downsampling{i + 1}")(x, B, Tn, lens32)
            d = self.pds_embed_dims[i]
            # Packed rows for the stage's layers where every kernel takes them (bf16, d = 256, heads of 64, enough rows): the
            # down-sampling convolution in front of the next stage reads padded rows again (zero padded frames: what its own
            # input mask makes of them, pdss2t_transformer.py:1100-1117)
            # (relative positions: the backward behind the skewed score gradient is s2t_relpos_glue — any length since round 5)
            pk = (Rows.ENABLED and dt == torch.bfloat16 and d == 256 and self.pds_attn_heads[i] * 64 == d and B * Tn >= Rows.MIN_ENC_ROWS
                  and Tn <= 65535 and not self.fusion_stages and (self.attn_type != "rel_pos" or Tn <= Fn._GLUE_MAX_T or not torch.is_grad_enabled())
                  # (weight gradients over packed rows: the 256 x 256 grouped kernel's operand rules, S2TTransformerEncoder._packed_ok)
                  and (not torch.is_grad_enabled() or Fn.wgrad256_eligible(
                      B * Tn, max(d * self.pds_ffn_ratios[i], 3 * d,
                                  Fn._pad8(self.ctc.ctc_projection.weight.shape[0]) if (i + 1 == self.pds_stages and getattr(self, "use_ctc", False)) else 0), d)))
            if pk:
                lens32 = Rows.attach(lens_memo[i], B, Tn, self._halo, tag=("pds", Fn.memo_owner(self), i))
                x = Rows.pack(x, lens32)
            c = Ctx(B, Tn, lens32, dt)
            if self.pds_position_embed[i]:
                if self.attn_type == "rel_pos":
                    c.pos_tab = TABLES.get("rel", Tn, d, x.device, dt)
                else:
                    tab = TABLES.get("sin", max(self.max_positions(), Tn) + 2, d, x.device)
                    x = AddPositions.apply(x, tab, lens32, Tn, 1.0)
            x = Fn.dropout(x, self.dropout_p if i == 0 else self.pds_dropout_p, self.training)  # :1118-1121
            for layer in getattr(self, f"stage{i + 1}"):
                x = layer(x, c, mask_output=False)
            if pk and i + 1 < self.pds_stages:
                x = Rows.unpack(x, lens32)
                lens32 = Rows.detached(lens32)
            elif pk:
                packed_last = lens32
            states.append((x, Tn, lens32))
        if self.fusion_stages:
            fused = None
            learned = isinstance(self.fusion_weight, torch.nn.Parameter)
            for k, i in enumerate(self.fusion_stages):
                # (a learned weight is one element of the fp32 parameter: autograd's product rule gives its gradient
                # <d fused, state_i> and accumulates it into the flat gradient buffer)
                wgt = self.fusion_weight[k].to(states[i][0].dtype) if learned else self.fusion_weight[k]
                s_, Ti, li = states[i]
                if self.fusion_mask:
                    s_ = MaskRows.apply(s_, li, Ti)
                if not self.fusion_no_prenorm:
                    s_ = getattr(self, f"fusion_pre_layer_norm{i + 1}")(s_)
                s_, To, _ = getattr(self, f"fusion_downsampling{i + 1}")(s_, B, Ti, li)
                assert To == Tn, "fusion branches must end at the last stage's frame rate"
                s_ = getattr(self, f"fusion_post_layer_norm{i + 1}")(s_)
                fused = wgt * s_ if fused is None else fused + wgt * s_  # three small elementwise passes (torch)
            x = fused
        if self.layer_norm is not None:
            x = self.layer_norm(x, rows=packed_last)
        lens = lens32.long()
        mask = torch.arange(Tn, device=x.device)[None, :] >= lens[:, None]
        ctc_logit = None
        if packed_last is not None:  # the last stage's rows stay packed for this package's consumers (s2t_transformer.py)
            rows_, dE = packed_last, self.embed_dim
            logit2d = self.ctc(x, out_dtype=self.ctc_out_dtype, rows=rows_) if self.use_ctc else None
            return {
                "encoder_out": Rows.LazyList([lambda: Rows.unpack(x, rows_).view(B, Tn, dE).transpose(0, 1)]),
                "ctc_logit": [] if logit2d is None else Rows.LazyList(
                    [lambda: Rows.unpack(logit2d.contiguous(), rows_).view(B, Tn, -1).transpose(0, 1)]),
                "inter_ctc_logits": [], "xctc_logit": [], "inter_xctc_logits": [],
                "encoder_padding_mask": [mask], "mixup": None, "encoder_embedding": [], "encoder_states": [],
                "src_tokens": [], "src_lengths": [],
                "packed": {"rows": rows_, "B": B, "T": Tn, "encoder_out": x, "ctc_logit": logit2d},
            }
        if self.use_ctc:
            ctc_logit = self.ctc(x, out_dtype=self.ctc_out_dtype).view(B, Tn, -1).transpose(0, 1)
        return {
            "encoder_out": [x.view(B, Tn, self.embed_dim).transpose(0, 1)],
            "ctc_logit": [] if ctc_logit is None else [ctc_logit],
            "inter_ctc_logits": [], "xctc_logit": [], "inter_xctc_logits": [],
            "encoder_padding_mask": [mask], "mixup": None, "encoder_embedding": [], "encoder_states": [],
            "src_tokens": [], "src_lengths": [],
        }

    reorder_encoder_out = None  # set below (shared implementation)


from .s2t_transformer import S2TTransformerEncoder as _Enc  # noqa: E402

PDSS2TTransformerEncoder.reorder_encoder_out = _Enc.reorder_encoder_out


@register_model("pdss2t_transformer")
class PDSS2TTransformerModel(S2TTransformerModel):
    """models/speech_to_text/pdss2t_transformer.py:147-288."""

    _REF_NAME = "pdss2t_transformer"

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)
        tgt = task.target_dictionary
        embed = Embedding(len(tgt), args.decoder_embed_dim, tgt.pad())
        return cls(PDSS2TTransformerEncoder(args, task, embed), TransformerDecoderScriptable(args, tgt, embed))


@register_model_architecture("pdss2t_transformer", "pdss2t_transformer")
def base_architecture(args):
    _d(args, "pds_stages", 4)
    _d(args, "pds_layers", "3_3_3_3")
    _d(args, "pds_ratios", "2_2_1_2")
    _d(args, "pds_ds_method", "conv")
    _d(args, "pds_embed_dims", "256_256_256_256")
    _d(args, "pds_embed_norm", False)
    _d(args, "pds_position_embed", "1_1_1_1")
    _d(args, "pds_attn_heads", "4_4_4_4")
    _d(args, "pds_ffn_ratios", "8_8_8_8")
    _d(args, "pds_kernel_sizes", "5_5_5_5")
    _d(args, "pds_fusion", False)
    _d(args, "pds_dropout", 0)
    _s2t_base(args)


@register_model_architecture("pdss2t_transformer", "pdss2t_transformer_s_8")
def pdss2t_transformer_s_8(args):
    """set_pds_base_8 (pdss2t_transformer.py:1496-1501) + pdss2t_transformer_s (:1561-1575)."""
    _d(args, "encoder_embed_dim", 256)
    _d(args, "pds_embed_norm", True)
    _d(args, "encoder_attention_heads", 4)
    _d(args, "decoder_attention_heads", 4)
    _d(args, "encoder_ffn_embed_dim", 256 * 8)
    base_architecture(args)
