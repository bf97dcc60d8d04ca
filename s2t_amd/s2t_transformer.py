"""S2T Transformer / Conformer encoder-decoder on the HIP path, with the reference's registration names,
constructor arguments, ``forward`` contracts and ``state_dict`` keys.

Reference: fairseq/models/speech_to_text/s2t_transformer.py (S2TTransformerModel :41-886, S2TTransformerEncoder
:887-2208, TransformerDecoderScriptable :2211-2253, architectures :2256-2470), fairseq/models/transformer.py
(TransformerDecoder :789-1515).  Out of scope here (raise NotImplementedError when asked for): mixup,
inter-CTC / XCTC / PAE, compression, DLCL history, layer-drop, quant-noise, adaptive softmax, incremental decoding.
"""
import math
from argparse import Namespace
from typing import Optional

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from . import rows as Rows
from .flat_params import FlatParameters
from .modules import (CTC, TABLES, Adapter, Conv1dSubsampling, Ctx, LayerNorm, Linear, MaskRows,
                      S2TTransformerEncoderLayer, TransformerDecoderLayer, pae_oracle_mask)
from .registry import model_base, reference_model_class, register_model, register_model_architecture

DEFAULT_MAX_SOURCE_POSITIONS = 6000
DEFAULT_MAX_TARGET_POSITIONS = 1024


def _unsupported(args, **flags):
    for name, off in flags.items():
        v = getattr(args, name, off)
        if v != off and v not in (None, False, 0, "", "none"):
            raise NotImplementedError("--%s=%r is outside the HIP hot path built so far" % (name.replace("_", "-"), v))


class _SinPosHolder(nn.Module):
    """Keeps the ``embed_positions._float_tensor`` buffer key of SinusoidalPositionalEmbedding."""

    def __init__(self):
        super().__init__()
        self.register_buffer("_float_tensor", torch.zeros(1))


class S2TTransformerEncoder(nn.Module):
    """Speech-to-text Transformer/Conformer encoder: Conv1d subsampler + N pre-LN layers (+ CTC head)."""

    def __init__(self, args, task=None, decoder_embed_tokens=None):
        super().__init__()
        _unsupported(args, inter_mixup=False, use_enc_dlcl=False, encoder_embed_linear=False,
                     layer_out_norm=False, encoder_layerdrop=0.0, inter_ctc_drop_prob=0, inter_ctc_mlo="")
        if not getattr(args, "disable_xctc", False):  # SATE sets it: XCTC then lives in the textual encoder (s2t_sate.py:842)
            _unsupported(args, inter_xctc_layers=None, xctc_weight=0)
        self.args = args
        d = args.encoder_embed_dim
        self.embed_dim = d
        self.padding_idx = 1
        self.embed_scale = 1.0 if args.encoder_no_scale_embedding else math.sqrt(d)
        self.dropout_p = float(args.dropout or 0.0)
        filters = [args.subsampling_filter] * (args.subsampling_layers - 1) + [d]
        self.subsample = Conv1dSubsampling(args.subsampling_layers, args.input_feat_per_channel * args.input_channels,
                                           filters, args.subsampling_kernel, args.subsampling_stride,
                                           args.subsampling_norm, args.subsampling_activation)
        self.attn_type = getattr(args, "encoder_attention_type", "selfattn")
        if self.attn_type != "rel_pos":
            self.embed_positions = _SinPosHolder()  # keeps the reference's `embed_positions._float_tensor` key
        self.embed_ln = LayerNorm(d) if getattr(args, "encoder_embed_norm", False) else None
        self.layer_padding_mask = bool(getattr(args, "layer_padding_mask", False))
        # rows kept behind every utterance of a packed batch: the reach of the conv module's depthwise kernel (s2t_amd/rows.py)
        self._halo = (int(getattr(args, "cnn_module_kernel", 31)) - 1) // 2 if getattr(args, "use_cnn_module", False) else 0
        self.layers = nn.ModuleList([S2TTransformerEncoderLayer(args) for _ in range(args.encoder_layers)])
        self.layer_norm = LayerNorm(d) if args.encoder_normalize_before else None
        self.use_ctc = getattr(args, "ctc_weight", 0) > 0
        if self.use_ctc:
            if getattr(args, "ctc_layer", 0) not in (0, args.encoder_layers):
                raise NotImplementedError("ctc_layer inside the stack")
            vocab = len(task.source_dictionary) if task is not None else args.vocab_size
            self.ctc = CTC(d, dictionary_size=vocab, dropout=args.dropout)
            if getattr(args, "share_ctc_and_embed", False) and decoder_embed_tokens is not None:
                self.ctc.ctc_projection.weight = decoder_embed_tokens.weight  # s2t_transformer.py:965-971
        # intermediate CTC heads (egs/mustc/asr/conf/inter.yaml; s2t_transformer.py:975-1100, forward :1881-1946):
        # after layer L (1-based; <= 0 counts from the top) a LayerNorm ``ctc_norm{L}`` (or the final one,
        # --share-inter-ctc-norm) feeds the shared top head (--share-inter-ctc) or the layer's own ``inter_ctc{L}``;
        # with --ctc-pae inter_league the layer output is then replaced by PAE(norm_x, logit) (``pae`` / ``pae{L}``)
        self.inter_ctc_layers = []
        self.ctc_pae_ground_truth_ratio = float(getattr(args, "ctc_pae_ground_truth_ratio", 0) or 0)
        self.pae_ground_truth_ratio = self.ctc_pae_ground_truth_ratio + float(
            getattr(args, "xctc_pae_ground_truth_ratio", 0) or 0)  # s2t_transformer.py:947-949 (read by CtcCriterion)
        spec = getattr(args, "inter_ctc_layers", None)
        if spec is not None and str(spec) not in ("", "None"):
            self.share_inter_ctc = bool(getattr(args, "share_inter_ctc", False))
            self.share_inter_ctc_norm = bool(getattr(args, "share_inter_ctc_norm", False))
            vocab = len(task.source_dictionary) if task is not None else args.vocab_size
            for t in str(spec).split(","):
                L = int(t)
                L = L + args.encoder_layers if L <= 0 else L
                if not self.share_inter_ctc_norm:
                    setattr(self, "ctc_norm%d" % L, LayerNorm(d))
                if not (self.use_ctc and self.share_inter_ctc):
                    head = CTC(d, dictionary_size=vocab, dropout=args.dropout)
                    if getattr(args, "share_ctc_and_embed", False) and decoder_embed_tokens is not None:
                        head.ctc_projection.weight = decoder_embed_tokens.weight
                    setattr(self, "inter_ctc%d" % L, head)
                self.inter_ctc_layers.append(L)
            self.pae_unnorm_input = bool(getattr(args, "pae_unnorm_input", False))
            self.pae_adaptive_gt = bool(getattr(args, "xctc_pae_ground_truth_ratio_adaptive", False))  # sic (:1129-1134)
            self.pae_gt_only_mistake = bool(getattr(args, "xctc_pae_ground_truth_only_mistake", False))
            _unsupported(args, share_pae_and_ctc=False, ctc_pae_ground_truth_ratio_decay=None)
            strategy = {"embed_norm": getattr(args, "pae_embed_norm", False), "out_norm": getattr(args, "pae_out_norm", False),
                        "ctc_temperature": getattr(args, "pae_ctc_temperature", 1.0), "gumbel": getattr(args, "pae_gumbel", False),
                        "distribution_hard": getattr(args, "pae_distribution_hard", None),
                        "gt_ratio": self.ctc_pae_ground_truth_ratio}
            pae_type = getattr(args, "ctc_pae", "none")
            if self.share_inter_ctc:
                self.pae = Adapter(d, pae_type, vocab, strategy=strategy)
            else:
                for L in self.inter_ctc_layers:
                    setattr(self, "pae%d" % L, Adapter(d, pae_type, vocab, strategy=strategy))
        # CTC-guided compression (egs/*/conf/dynamic.yaml; ctor :1301-1359, forward :1948-2040): after the intermediate
        # CTC head of a listed layer, frames whose blank posterior reaches the threshold are dropped (``create``: every
        # utterance left-packed, T' shrinks to the longest), then ``compression_norm{L}`` and fresh positions
        self.compression_layers = []
        spec = getattr(args, "compression_layers", None)
        if spec is not None and str(spec) not in ("", "None", "none"):
            if getattr(args, "compression_metric", "ratio") != "threshold":
                raise NotImplementedError("--compression-metric %s" % getattr(args, "compression_metric", "ratio"))
            if getattr(args, "compression_mode", "create") != "create":
                raise NotImplementedError("--compression-mode mask (key masks with holes; the attention kernels take lengths)")
            self.compression_layers = [int(t) for t in str(spec).split(",")]
            assert all(L in self.inter_ctc_layers for L in self.compression_layers), "compression needs an inter-CTC head"
            thr = [float(t) for t in str(getattr(args, "compression_threshold", "1.0")).split(",")]
            assert len(thr) in (1, len(self.compression_layers))
            thr = thr if len(thr) == len(self.compression_layers) else thr * len(self.compression_layers)
            self.compression_thresholds = dict(zip(self.compression_layers, thr))
            self.compression_pos = bool(getattr(args, "compression_pos", False))
            self.compression_norm = bool(getattr(args, "compression_norm", False))
            if self.compression_pos and self.attn_type != "rel_pos":
                self.compression_embed_positions = _SinPosHolder()
            if self.compression_norm:
                for L in self.compression_layers:
                    setattr(self, "compression_norm%d" % L, LayerNorm(d))
        self.compute_dtype = torch.float32
        self.ctc_out_dtype = None  # None -> compute dtype; eval sets fp32 (bit-exact greedy wants fp32 logits)
        self.num_updates = 0

    # -- fairseq encoder protocol ------------------------------------------------------------------
    def max_positions(self):
        return getattr(self.args, "max_source_positions", DEFAULT_MAX_SOURCE_POSITIONS)

    def set_num_updates(self, n):
        self.num_updates = n

    def set_ctc_infer(self, ctc_infer, post_process, src_dict=None, tgt_dict=None, path=None):
        if hasattr(self, "ctc"):
            self.ctc.set_infer(ctc_infer, post_process, src_dict, path)

    def set_flag(self, **kwargs):
        pass

    def dump(self, fstream, info=""):
        pass

    def _packed_ok(self, dt, B, Tp):
        """Packed rows where every kernel on the path takes them: bf16, heads of 64, enough rows.  At the recipes' width
        (d = 256) the fused row-block kernels carry training and inference; at other widths (the NAST recipe's d = 512) the
        GEMM / LayerNorm composition takes the row map too — in INFERENCE: the convolution module's backward exists for packed
        rows on the fused d = 256 kernel only.  CTC-guided compression rewrites the frame axis between the layers: padded."""
        if not Rows.ENABLED or dt != torch.bfloat16:
            return False
        d = self.embed_dim
        if d != 256 and torch.is_grad_enabled():
            return False
        h = getattr(self.layers[0].self_attn, "num_heads", None) or getattr(self.layers[0].self_attn, "h", 0)
        if h * 64 != d or self.compression_layers:
            return False
        if self.inter_ctc_layers and (self.pae_ground_truth_ratio > 0 and self.training
                                      or getattr(self, "decode_inter_logits", False)):
            return False  # (intermediate heads run packed unless the ground-truth curriculum is on — both of the criterion's
                          #  passes then read (T, B, V) views and mix (B, T) oracle labels in — or a decoder reads an intermediate
                          #  head: --ctc-inter-logit)
        if self.attn_type == "rel_pos" and Tp > Fn._GLUE_MAX_T and torch.is_grad_enabled():
            return False  # (S2T_GLUE_MAX_T: the round-4 routing of the relative-position backward, padded rows only)
        if torch.is_grad_enabled():
            # weight gradients over packed rows exist on the 256 x 256 grouped kernel only (it reads the live row count on the
            # device): its operand rules must hold for the widest operands of this stack — the feed-forward hidden activation and
            # the CTC logits — or the batch stays padded (S2T_WG_256=0, a vocabulary beyond 2 GiB of logits)
            widest = max(int(self.args.encoder_ffn_embed_dim), 3 * d,
                         Fn._pad8(self.ctc.ctc_projection.weight.shape[0]) if self.use_ctc else 0)
            if not Fn.wgrad256_eligible(B * Tp, widest, d):
                return False
        return B * Tp >= Rows.MIN_ENC_ROWS and Tp <= 65535

    # -- forward -------------------------------------------------------------------------------------
    def forward(self, src_tokens, src_lengths=None, **kwargs):
        """src_tokens (B, T, C) float, src_lengths (B,) long -> dict of lists (s2t_transformer.py:2142-2154)."""
        if not src_tokens.is_cuda:
            raise RuntimeError("s2t_amd runs on the GPU only; there is no CPU fallback")
        dt = self.compute_dtype
        B, T, _ = src_tokens.shape
        d = self.embed_dim
        T1 = (T - 1) // 2 + 1
        Tp = (T1 - 1) // 2 + 1
        if src_lengths is None:
            src_lengths = torch.full((B,), T, dtype=torch.long, device=src_tokens.device)
        def length_bookkeeping(sl):
            ln = self.subsample.get_out_seq_lens_tensor(sl)
            return ln, ln.to(torch.int32), torch.arange(Tp, device=sl.device)[None, :] >= ln[:, None]

        def length_bookkeeping_into(sl, outs):  # the same into the memo's tensors, one launch (functional._recompute_in_place)
            K.subsampled_lengths(sl, Tp, outs[0], outs[1], outs[2])

        if src_lengths.dtype == torch.int64 and src_lengths.is_contiguous():
            length_bookkeeping.into = length_bookkeeping_into
        lens, lens32, encoder_padding_mask = Fn.batch_memo(("enc_lens", Fn.memo_owner(self), Tp), (src_lengths,), length_bookkeeping)
        x = self.subsample(src_tokens, Rows.detached(lens32), dt)  # [B*T', d], padded frames zeroed (:1765)
        if self._packed_ok(dt, B, Tp):
            # Packed rows (s2t_amd/rows.py): from here to the end of the encoder only the frames (and the conv module's halo
            # rows) are computed; the buffers keep their B * T' rows.  ``lens32`` carries the geometry to every launch.
            lens32 = Rows.attach(lens32, B, Tp, self._halo, tag=("enc", Fn.memo_owner(self)))
            x = Rows.pack(x, lens32)
        else:
            lens32 = Rows.detached(lens32)
        c = Ctx(B, Tp, lens32, dt)
        if self.embed_ln is not None:
            x = self.embed_ln(x, rows=c.rows)  # :1769
        if self.attn_type == "rel_pos":
            c.pos_tab = TABLES.get("rel", Tp, d, x.device, dt)  # not added to x (:1777-1778)
            if self.embed_scale != 1.0:
                x = x * self.embed_scale
        else:
            tab = TABLES.get("sin", max(self.max_positions(), Tp) + 2, d, x.device)
            x = AddPositions.apply(x, tab, lens32, Tp, self.embed_scale)  # :1773-1787
        x = Fn.dropout(x, self.dropout_p, self.training)  # dropout_module (:1794)
        if self.layer_padding_mask and Rows.K.rows_geom(lens32) is None:
            # layer 0's masked_fill (:1828-1836); later layers: fused in final_norm.  Packed rows hold no padded frames and their
            # halo rows are zero from Rows.pack on (dropout keeps zeros), and PackFn.backward drops the halo rows' gradient
            x = MaskRows.apply(x, lens32, Tp)
        n = len(self.layers)
        inter_ctc_logits, inter_packed = [], []
        ctc_orc = ctc_force_emit = None
        # gradient stages (Fn.grad_stage, data-parallel steps only): where the weight gradients queued so far run and their
        # buckets start reducing beside the rest of backward (Fn.GRAD_STAGES)
        stage_at = Fn.grad_stage_layers(n)
        if self.training:  # nn.BatchNorm1d's num_batches_tracked of every conv module: one launch instead of one per layer
            ctrs = [l.conv_module.norm.num_batches_tracked for l in self.layers if getattr(l, "conv_module", None) is not None]
            if ctrs:
                for l in self.layers:
                    if getattr(l, "conv_module", None) is not None:
                        l.conv_module.counter_elsewhere = True
                torch._foreach_add_(ctrs, 1)
        pos_all = None
        if self.attn_type == "rel_pos":  # the position table is projected for all layers by one batched launch
            pos_all = Fn.project_positions(c.pos_tab, [l.self_attn.linear_pos.weight for l in self.layers])
        for i, layer in enumerate(self.layers):
            tap = (i + 1) in self.inter_ctc_layers  # the head reads the layer output BEFORE the next layer's mask
            c.cur_pos_p = pos_all[i] if (pos_all is not None and c.pos_tab is not None and pos_all[i].shape[0] == c.pos_tab.shape[0]) else None
            if i in stage_at:
                x = Fn.grad_stage(x)
            x = layer(x, c, mask_output=self.layer_padding_mask and i + 1 < n and not tap)
            if tap:
                L = i + 1
                norm = self.layer_norm if self.share_inter_ctc_norm else getattr(self, "ctc_norm%d" % L)
                head = self.ctc if (self.use_ctc and self.share_inter_ctc) else getattr(self, "inter_ctc%d" % L)
                pae = self.pae if self.share_inter_ctc else getattr(self, "pae%d" % L)
                rows_t = c.rows  # (packed rows: the tap's LayerNorm, head and PAE run on the frames only)
                norm_x = norm(x, rows=rows_t)
                # (an INTERMEDIATE head's logits feed the PAE softmax and the training losses: compute dtype, as in training;
                # ctc_out_dtype = fp32 is for the logits that are decoded — CTCDecoder(--ctc-inter-logit k) decodes from an
                # intermediate head and sets ``decode_inter_logits``: they then follow ctc_out_dtype)
                logit2d = head(norm_x, out_dtype=self.ctc_out_dtype if (pae.adapter_type == "none" or getattr(
                    self, "decode_inter_logits", False)) else None, rows=rows_t)
                if rows_t is not None:
                    # the (T, B, V) view is unpacked only if somebody reads it; this package's criterion takes the packed rows
                    inter_packed.append(logit2d)
                    inter_logit = Rows.LazyList([
                        (lambda l2=logit2d, r=rows_t, T_=Tp: Rows.unpack(l2.contiguous(), r).view(B, T_, -1).transpose(0, 1)),
                        (lambda m=encoder_padding_mask: m)])
                    il = None
                else:
                    il = logit2d.view(B, Tp, -1).transpose(0, 1)
                    inter_logit = [il, encoder_padding_mask]  # the reference's [logit, padding mask] pairs
                orc = msk = None
                if self.ctc_pae_ground_truth_ratio > 0:  # :1904-1935
                    oracle = (kwargs.get("ctc_alignment_oracle") or {}).get("ctc")
                    if oracle is not None:
                        if ctc_orc is None:
                            ctc_orc = pae_oracle_mask(oracle, self.ctc_pae_ground_truth_ratio, self.pae_adaptive_gt,
                                                      self.pae_gt_only_mistake, (kwargs.get("pae_oracle_masks") or {}).get("ctc"))
                            ctc_force_emit = ctc_orc[2]
                        orc, msk = ctc_orc[0], ctc_orc[1]
                        inter_logit = [il, None, ctc_force_emit]
                if pae.adapter_type != "none":
                    x = pae(x if self.pae_unnorm_input else norm_x, logit2d, orc, msk, rows=rows_t)
                inter_ctc_logits.append(inter_logit)
                if L in self.compression_layers:
                    x, lens32, Tp, encoder_padding_mask = self._compress(x, logit2d, L, B, Tp, lens32, encoder_padding_mask)
                    c = Ctx(B, Tp, lens32, dt)
                    if self.attn_type == "rel_pos":
                        c.pos_tab = TABLES.get("rel", Tp, d, x.device, dt)  # :1838-1843: positions follow the new length
                elif self.layer_padding_mask and i + 1 < n:
                    x = MaskRows.apply(x, lens32, Tp)
        if self.layer_norm is not None:
            x = self.layer_norm(x, rows=c.rows)
        if Fn.GRAD_STAGES >= 4:
            x = Fn.grad_stage(x)  # everything behind the encoder output (decoder, CTC head) forms the first gradient stage
        ctc_logit = None
        packed = None
        if c.rows is not None:
            # the reference's T x B x C tensors are materialised only if somebody reads them (Rows.LazyList: zero-filled padded
            # frames, as the reference's masked tensors hold); the consumers of this package take the packed rows under "packed"
            rows_ = c.rows
            greedy = None
            if (self.use_ctc and getattr(self, "ctc_greedy_only", False) and self.ctc_out_dtype == torch.float32
                    and self.ctc.greedy_supported(x)):
                # CTCDecoder.generate is the only reader and wants arg-max + its log-probability per frame: the head and the
                # arg-max in one launch, the [rows, V] fp32 logits never stored (csrc/ctc_head.hip); whoever still asks for
                # "ctc_logit" gets it computed on demand
                greedy = self.ctc.greedy(x, rows=rows_)
                logit2d = None
                ctc_list = Rows.LazyList([lambda: Rows.unpack(self.ctc(x, out_dtype=self.ctc_out_dtype, rows=rows_).contiguous(), rows_)
                                          .view(B, Tp, -1).transpose(0, 1)])
            else:
                logit2d = self.ctc(x, out_dtype=self.ctc_out_dtype, rows=rows_) if self.use_ctc else None
                ctc_list = [] if logit2d is None else Rows.LazyList(
                    [lambda: Rows.unpack(logit2d.contiguous(), rows_).view(B, Tp, -1).transpose(0, 1)])
            packed = {"rows": rows_, "B": B, "T": Tp, "encoder_out": x, "ctc_logit": logit2d, "inter_ctc_logit": inter_packed,
                      "ctc_greedy": greedy}
            enc_list = Rows.LazyList([lambda: Rows.unpack(x, rows_).view(B, Tp, d).transpose(0, 1)])
            return {
                "encoder_out": enc_list, "ctc_logit": ctc_list, "inter_ctc_logits": inter_ctc_logits, "xctc_logit": [],
                "inter_xctc_logits": [], "encoder_padding_mask": [encoder_padding_mask], "mixup": None,
                "encoder_embedding": [], "encoder_states": [], "src_tokens": [], "src_lengths": [], "packed": packed,
            }
        if self.use_ctc:
            logit2d = self.ctc(x, out_dtype=self.ctc_out_dtype)
            ctc_logit = logit2d.view(B, Tp, -1).transpose(0, 1)
            if ctc_force_emit is not None:
                ctc_logit = [ctc_logit, None, ctc_force_emit]  # :2137-2138
        return {
            "encoder_out": [x.view(B, Tp, d).transpose(0, 1)],  # T x B x C (view of the batch-major buffer)
            "ctc_logit": [] if ctc_logit is None else [ctc_logit],
            "inter_ctc_logits": inter_ctc_logits,
            "xctc_logit": [],
            "inter_xctc_logits": [],
            "encoder_padding_mask": [encoder_padding_mask],
            "mixup": None,
            "encoder_embedding": [],
            "encoder_states": [],
            "src_tokens": [],
            "src_lengths": [],
        }

    def _compress(self, x, logit2d, L, B, T, lens32, mask):
        """s2t_transformer.py:1948-2040 for layer L.  Eager form: one D2H copy of the B new lengths decides the new T' (the
        reference synchronises on ``max(keep_flag.sum(0))`` as well).  ``compression_bounded`` (set by Trainer.capture, implied
        while a stream is capturing): no host copy, T stays the bound."""
        src, new_lens = Fn.ctc_compress_plan(logit2d.detach(), lens32, B, T, 0, self.compression_thresholds[L])
        # (``compression_bounded``: True = always; "train" = the captured TRAINING step's form, set by Trainer.capture and cleared
        # by release_trainer — eval and generation on the same model keep the exact form and the reference's shapes)
        cb = getattr(self, "compression_bounded", False)
        if cb is True or (cb == "train" and self.training) or (x.is_cuda and torch.cuda.is_current_stream_capturing()):
            # Capturable form: the frame axis keeps its uncompressed BOUND T and only the lengths change, on the device.  The
            # reference's two host decisions (:1996-2001: compress only when no utterance would become empty and something is
            # dropped) select between the plan and the identity with device-side flags.  Outputs equal the exact form's on the
            # first max(new_lens) frames and are padding behind (masks / lengths say so to every consumer).
            ok = (new_lens.min() > 0) & (new_lens.sum() != B * T)
            ident = torch.arange(T, dtype=torch.int32, device=x.device)[None, :].expand(B, T)
            src = torch.where(ok, src, ident).contiguous()
            new_lens = torch.where(ok, new_lens, lens32)
            x = Fn.CompressRowsFn.apply(x, src, new_lens, B, T, T)
            lens32 = new_lens
            mask = torch.arange(T, device=x.device)[None, :] >= lens32[:, None]
        else:
            nl = new_lens.cpu()
            kept_all = int(nl.sum()) == B * T  # keep_flag.all(): padded frames count as dropped
            if int(nl.min()) > 0 and not kept_all:
                Tn = int(nl.max())
                x = Fn.CompressRowsFn.apply(x, src, new_lens, B, T, Tn)
                T, lens32 = Tn, new_lens
                mask = torch.arange(T, device=x.device)[None, :] >= lens32[:, None]
        if self.compression_norm:
            x = getattr(self, "compression_norm%d" % L)(x)
        if self.compression_pos and self.attn_type != "rel_pos":
            tab = TABLES.get("sin", max(self.max_positions(), T) + 2, self.embed_dim, x.device)
            x = AddPositions.apply(x, tab, lens32, T, 1.0)
        x = MaskRows.apply(x, lens32, T)  # :2037-2040
        return x, lens32, T, mask

    def reorder_encoder_out(self, encoder_out, new_order):
        """s2t_transformer.py:2156-2208."""
        def sel(key, dim):
            return [t.index_select(dim, new_order) for t in encoder_out.get(key, []) if t is not None]

        return {
            "encoder_out": sel("encoder_out", 1),
            "ctc_logit": sel("ctc_logit", 1),
            "xctc_logit": sel("xctc_logit", 1),
            "encoder_padding_mask": sel("encoder_padding_mask", 0),
            "encoder_embedding": sel("encoder_embedding", 0),
            "encoder_states": [s.index_select(1, new_order) for s in encoder_out.get("encoder_states", [])],
            "src_tokens": [],
            "src_lengths": [],
        }


class AddPositions(torch.autograd.Function):
    """x = scale*x + sinusoid(position of non-pad frame) (s2t_transformer.py:1773-1787; positions start at 2)."""

    @staticmethod
    def forward(ctx, x, tab, lens, T, scale):
        from . import kernels as K

        y = x.clone()
        K.add_positions(y, tab, lens, y.shape[0], T, y.shape[1], scale, 2)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        return (dy if ctx.scale == 1.0 else dy * ctx.scale), None, None, None, None


class Embedding(nn.Module):
    """models/transformer.py:1518-1522 — N(0, d^-0.5), pad row zero."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight, mean=0, std=embedding_dim ** -0.5)
        with torch.no_grad():
            self.weight[padding_idx].fill_(0)
        self.padding_idx = padding_idx
        self.embedding_dim = embedding_dim


class TransformerDecoderScriptable(nn.Module):
    """Teacher-forced cross-attention decoder (models/transformer.py:802-1448, s2t_transformer.py:2211-2253)."""

    def __init__(self, args, dictionary, embed_tokens):
        super().__init__()
        _unsupported(args, use_dec_dlcl=False, decoder_layerdrop=0.0, adaptive_softmax_cutoff=None,
                     layernorm_embedding=False, decoder_learned_pos=False, no_token_positional_embeddings=False)
        self.args = args
        self.dictionary = dictionary
        d = args.decoder_embed_dim
        self.embed_dim = d
        self.padding_idx = embed_tokens.padding_idx
        self.embed_tokens = embed_tokens
        self.embed_scale = 1.0 if getattr(args, "no_scale_embedding", False) else math.sqrt(d)
        self.embed_positions = _SinPosHolder()
        self.register_buffer("version", torch.tensor([3.0]))
        self.layers = nn.ModuleList([TransformerDecoderLayer(args) for _ in range(args.decoder_layers)])
        self.layer_norm = LayerNorm(d) if args.decoder_normalize_before else None
        self.output_projection = Linear(d, len(dictionary), bias=False)
        if args.share_decoder_input_output_embed:
            self.output_projection.weight = self.embed_tokens.weight
        else:
            nn.init.normal_(self.output_projection.weight, mean=0, std=d ** -0.5)
        self.compute_dtype = torch.float32
        self.logits_dtype = None

    def max_positions(self):
        return getattr(self.args, "max_target_positions", DEFAULT_MAX_TARGET_POSITIONS)

    def extract_features(self, prev_output_tokens, encoder_out=None, incremental_state=None, packed_out=False, **unused):
        """``packed_out`` (this package's criterion): where the target rows run packed (s2t_amd/rows.py) they are returned as they
        are — ``[B * U, d]`` with ``extra["packed"]`` naming their geometry — instead of the reference's ``B x U x d``."""
        if incremental_state is not None:
            return self._extract_features_incremental(prev_output_tokens, encoder_out, incremental_state)
        B, U = prev_output_tokens.shape
        d = self.embed_dim
        dev = prev_output_tokens.device
        pad_idx = self.padding_idx

        def token_bookkeeping(tok):
            nonpad = tok.ne(pad_idx)
            pos = (torch.cumsum(nonpad, dim=1) * nonpad + pad_idx).to(torch.int32)  # utils.py:240-250
            # key-padding of the target side: pads sit at the end for left-aligned targets; general masks
            # (pads in the middle) would need a mask tensor, the collater never produces them
            return tok.contiguous(), pos.contiguous(), nonpad.sum(1).to(torch.int32)

        def token_bookkeeping_into(tok, outs):  # the same into the memo's tensors, one launch
            if outs[0] is not tok:
                outs[0].copy_(tok)
            K.token_positions(outs[0], pad_idx, outs[1], outs[2])

        if prev_output_tokens.dtype == torch.int64:
            token_bookkeeping.into = token_bookkeeping_into
        tok, pos, self_lens = Fn.batch_memo(("dec_tokens", Fn.memo_owner(self)), (prev_output_tokens,), token_bookkeeping)
        tab = TABLES.get("sin", self.max_positions() + self.padding_idx + 1, d, dev)
        x = Fn.embedding(tok, pos, self.embed_tokens.weight, tab, self.embed_scale, self.padding_idx)
        x = Fn.dropout(x, float(self.args.dropout or 0.0), self.training)  # dropout_module (transformer.py:1328)
        # Packed target rows: the collater left-aligns the targets and pads them to the longest of the batch
        # (data/audio/speech_to_text_dataset.py:445-470); no decoder module looks across a target's end, so there is no halo.
        dec_rows = None
        widest = max(int(self.args.decoder_ffn_embed_dim), 3 * d, Fn._pad8(self.output_projection.weight.shape[0]))
        if (Rows.ENABLED and x.dtype == torch.bfloat16 and d == 256 and self.layers[0].self_attn.num_heads * 64 == d
                and B * U >= Rows.MIN_DEC_ROWS and U <= 65535
                and (not torch.is_grad_enabled() or Fn.wgrad256_eligible(B * U, widest, d))):  # (see _packed_ok)
            dec_rows = self_lens = Rows.attach(self_lens, B, U, 0, tag=("dec", Fn.memo_owner(self)))
            x = Rows.pack(x, dec_rows)
        pk = encoder_out.get("packed")
        if (pk is not None and pk["B"] == B and Fn._use_fused_attention(pk["encoder_out"].dtype, d // self.layers[0].encoder_attn.num_heads)
                and (not torch.is_grad_enabled() or Fn.wgrad256_eligible(B * pk["T"], 2 * d * len(self.layers), d))):
            # the encoder's packed rows (s2t_amd/rows.py) are the memory as they are: the key side of every encoder-decoder
            # attention reads utterance b's frames from row cu[b]
            mem, Tm, mem_lens = pk["encoder_out"], pk["T"], pk["rows"]
        else:
            mem_tbc = encoder_out["encoder_out"][0]
            Tm = mem_tbc.shape[0]
            mem = mem_tbc.transpose(0, 1).contiguous().view(B * Tm, d)
            (mem_lens,) = Fn.batch_memo(("dec_mem_lens", Fn.memo_owner(self)), (encoder_out["encoder_padding_mask"][0],),
                                        lambda m: ((~m).sum(1).to(torch.int32),))
        # reference (:1340-1342): pad KEYS are masked for every query; pad queries still attend
        # k | v of the encoder memory for all layers' encoder-decoder attention in one projection (Fn.cross_kv)
        L = len(self.layers)
        ckv = Fn.cross_kv(mem, [layer.encoder_attn._prm() for layer in self.layers], self.layers[0].encoder_attn.num_heads,
                          rows=mem_lens)
        for i, layer in enumerate(self.layers):
            x = layer(x, mem, B, U, Tm, self_lens, mem_lens, mem_kv=(ckv[0], i, L, ckv[1]) if ckv is not None else None)
        if self.layer_norm is not None:
            x = self.layer_norm(x, rows=dec_rows)
        extra = {"attn": [None], "inner_states": [], "mixup": None}
        if dec_rows is not None:
            if packed_out:
                extra["packed"] = {"rows": dec_rows, "B": B, "U": U}
                return x, extra
            x = Rows.unpack(x, dec_rows)  # the reference's layout (zero rows at the padded positions)
        return x.view(B, U, d), extra

    def _extract_features_incremental(self, prev_output_tokens, encoder_out, incremental_state):
        """models/transformer.py:1290-1312 with ``incremental_state``: only the last position is embedded and pushed
        through the layers; self-attention keys/values of earlier positions come from the per-layer caches kept in
        ``incremental_state`` (modules/multihead_attention.py:302-339), the projected encoder memory is cached too."""
        if torch.is_grad_enabled():
            raise RuntimeError("incremental decoding is inference only: wrap the generator in torch.no_grad()")
        Bb, U = prev_output_tokens.shape
        d = self.embed_dim
        dev = prev_output_tokens.device
        nonpad = prev_output_tokens.ne(self.padding_idx)
        pos = (torch.cumsum(nonpad, dim=1) * nonpad + self.padding_idx).to(torch.int32)[:, -1:].contiguous()
        tab = TABLES.get("sin", self.max_positions() + self.padding_idx + 1, d, dev)
        x = Fn.embedding(prev_output_tokens[:, -1:].contiguous(), pos, self.embed_tokens.weight, tab, self.embed_scale,
                         self.padding_idx)
        mem_tbc = encoder_out["encoder_out"][0]
        Tm = mem_tbc.shape[0]
        mem_lens = (~encoder_out["encoder_padding_mask"][0]).sum(1).to(torch.int32)
        state = incremental_state.setdefault("s2t_amd.decoder", {"layers": [dict() for _ in self.layers]})
        mem = None
        if "mem_kv" not in state["layers"][0]:
            mem = mem_tbc.transpose(0, 1).contiguous().view(Bb * Tm, d)
        for layer, st in zip(self.layers, state["layers"]):
            x = layer.step(x, st, U, mem, Bb, Tm, mem_lens)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        return x.view(Bb, 1, d), {"attn": [None], "inner_states": [], "mixup": None}

    def reorder_incremental_state(self, incremental_state, new_order):
        """modules/multihead_attention.py:574-592 / fairseq_incremental_decoder.py: caches follow the surviving beams."""
        state = incremental_state.get("s2t_amd.decoder")
        if state is None:
            return
        for st in state["layers"]:
            for k in list(st.keys()):
                st[k] = st[k].index_select(0, new_order)

    reorder_incremental_state_scripting = reorder_incremental_state

    def output_layer(self, features, rows=None):
        if features.dim() == 2:  # packed target rows: logits stay [B * U, V]
            return self.output_projection(features, out_dtype=self.logits_dtype, rows=rows)
        B, U, d = features.shape
        y = self.output_projection(features.reshape(B * U, d), out_dtype=self.logits_dtype)
        return y.view(B, U, -1)

    def forward(self, prev_output_tokens, encoder_out=None, incremental_state=None, features_only=False, packed_out=False,
                **unused):
        x, extra = self.extract_features(prev_output_tokens, encoder_out, incremental_state, packed_out=packed_out)
        if not features_only:
            x = self.output_layer(x, rows=extra["packed"]["rows"] if "packed" in extra else None)
        return x, extra

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        logits = net_output[0].float()
        return torch.log_softmax(logits, dim=-1) if log_probs else torch.softmax(logits, dim=-1)


class _DictLike:
    """Minimal stand-in for fairseq.data.Dictionary when fairseq is absent: bos=0 (CTC blank), pad=1, eos=2, unk=3."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def pad(self):
        return 1

    def eos(self):
        return 2

    def bos(self):
        return 0

    def unk(self):
        return 3


class FakeTask:
    def __init__(self, vocab):
        self.source_dictionary = self.target_dictionary = _DictLike(vocab)

    def get_source_dictionary(self, i):
        return self.source_dictionary


class _HipModel(model_base()):
    """Common base of the HIP models: ``BaseFairseqModel`` under fairseq (so that its registry, checkpoint utilities and
    trainer accept the class), ``nn.Module`` otherwise.

    ``prepare`` flattens the parameters (fp32 master + gradient buffer + bf16 shadow, flat_params.py).  Callers of the
    bundled harness call it themselves; under the reference's trainer nothing does, so the usual hooks fold it in:
    ``model.bfloat16()`` / ``.half()`` (fairseq/trainer.py:85-90) only RECORD the compute dtype — the master weights stay
    fp32, the reference's FP16Optimizer keeps fp32 masters too (optim/fp16_optimizer.py:30-60) — and the first forward
    through the model prepares on the device the parameters were moved to.  Before every forward the gradient views are
    re-attached when an optimizer dropped them and the bf16 shadow is refreshed when the master weights changed."""

    flat: Optional[FlatParameters] = None
    _compute_dtype = None
    shadow_managed = False  # True: the owner of the optimizer step (s2t_amd.trainer.Trainer) refreshes the bf16 shadow itself
    _shadow_dirty = True
    _step_notified = False  # attach_optimizer / notify_optimizer_step in use: refresh exactly when told
    _refreshed_at = -1      # functional.backward_count() at the last heuristic refresh

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.register_forward_pre_hook(_HipModel._before_forward)

    def __setattr__(self, name, value):
        """The criteria enter through ``model.encoder(...)`` / ``model.decoder(...)``, not ``model(...)`` — the reference's do
        too (criterions/label_smoothed_cross_entropy_with_ctc.py:80-95) — so the before-forward work hangs on those entry
        modules as well."""
        super().__setattr__(name, value)
        if name in ("encoder", "decoder") and isinstance(value, nn.Module) and not getattr(value, "_s2t_entry_hook", False):
            import weakref

            value._s2t_entry_hook = True
            owner = weakref.ref(self)
            value.register_forward_pre_hook(lambda mod, args: _HipModel._before_forward(owner(), args) if owner() is not None else None)

    def prepare(self, dtype=torch.float32, device="cuda"):
        """Move to the GPU, flatten parameters (fp32 master + grads + bf16 shadow) and set the compute dtype."""
        nn.Module.to(self, device)
        self.flat = FlatParameters(self, dtype)
        self._compute_dtype = dtype
        for m in self.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype
        return self

    def bfloat16(self):
        self._compute_dtype = torch.bfloat16
        return self

    def half(self):
        self._compute_dtype = torch.bfloat16  # the HIP path's reduced precision is bf16 (fp32 accumulate and statistics)
        return self

    @staticmethod
    def _before_forward(self, args):
        if self.flat is None:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("s2t_amd models run on the GPU only: move the model with .cuda() / .to(device) first")
            self.prepare(self._compute_dtype or torch.float32, dev)
        else:
            self.flat.reattach_grads()
            # The bf16 shadow (and the transposed copies made from it) must follow the fp32 masters.  The bundled Trainer's
            # s2t_adam_step rewrites the shadow itself (``shadow_managed``).  ANY other optimizer — torch.optim, fairseq's Adam
            # under the reference trainer — steps ``p`` / ``p.data`` in place, which no version counter of the flat buffer
            # records (each Parameter view has its own counter and ``.data`` aliases none), so the shadow is rewritten before
            # every forward: one cast launch over the flat buffer.
            # every forward that can be followed by an optimizer step (training mode, autograd on), and the first forward after
            # a train() / eval() switch or a checkpoint load; inference loops keep the shadow they have (one 56 us cast of
            # the 30 M parameters per forward otherwise)
            # With ``attach_optimizer`` (or explicit ``notify_optimizer_step`` calls) the refresh follows the optimizer's steps
            # exactly — whatever mode the model is in — and the training-mode heuristic is off.  The hook sits on the model and on
            # its encoder / decoder (a criterion may enter through either): one refresh per step, not one per entry.
            heuristic = (self.training and torch.is_grad_enabled() and not self._step_notified
                         and self._refreshed_at != Fn.backward_count())
            if self.flat.shadow is not None and not self.shadow_managed and (self._shadow_dirty or heuristic):
                self.flat.refresh_shadow()
                self.flat.mark_transposed_stale()
                self._shadow_dirty = False
                self._refreshed_at = Fn.backward_count()  # (an optimizer step follows a backward pass: once per pass)
        return None

    def notify_optimizer_step(self):
        """Tell the model that an optimizer other than the bundled Trainer's has stepped the fp32 masters: the bf16 shadow (and
        the transposed weight copies) are rewritten before the next forward, in training or eval mode, with or without autograd.
        Once called, the per-forward heuristic (refresh before every training forward) is switched off."""
        self._shadow_dirty = True
        self._step_notified = True

    def attach_optimizer(self, optimizer):
        """``torch.optim.Optimizer.register_step_post_hook`` -> ``notify_optimizer_step`` (fairseq optimizers: pass
        ``optimizer.optimizer``); returns the hook handle."""
        import weakref

        me = weakref.ref(self)
        return optimizer.register_step_post_hook(lambda *a, **k: me() is not None and me().notify_optimizer_step())

    def release_trainer(self):
        """The bundled Trainer no longer updates this model (it set ``shadow_managed``): back to refreshing the shadow here."""
        self.shadow_managed = False
        self._shadow_dirty = True
        for mod in self.modules():  # the captured step's bounded compression form (Trainer.capture)
            if getattr(mod, "compression_bounded", False) == "train":
                mod.compression_bounded = False

    def train(self, mode=True):
        self._shadow_dirty = True  # whoever trained may have stepped the masters since the last forward
        return super().train(mode)

    def load_state_dict(self, state_dict, strict=True, model_cfg=None, args=None, **kw):
        r = nn.Module.load_state_dict(self, state_dict, strict=strict, **kw)
        if self.flat is not None:
            self.flat.refresh_shadow()
            self.flat.mark_transposed_stale()
            self._shadow_dirty = False
        return r


@register_model("s2t_transformer")
class S2TTransformerModel(_HipModel):
    """models/speech_to_text/s2t_transformer.py:41-886 — encoder + decoder, ``build_model(args, task)``."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        self.flat: Optional[FlatParameters] = None

    _REF_NAME = "s2t_transformer"

    @classmethod
    def add_args(cls, parser):
        """The command-line flags of the recipes are the reference's (s2t_transformer.py:60-800): under fairseq they are
        taken from the reference class registered under the same name, so every recipe flag parses; without fairseq
        the harness builds ``args`` namespaces directly (``recipe_args``)."""
        ref = reference_model_class(cls._REF_NAME)
        if ref is not None:
            ref.add_args(parser)

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)
        tgt = task.target_dictionary
        embed = Embedding(len(tgt), args.decoder_embed_dim, tgt.pad())
        encoder = S2TTransformerEncoder(args, task, embed)
        decoder = TransformerDecoderScriptable(args, tgt, embed)
        return cls(encoder, decoder)

    def forward(self, src_tokens, src_lengths, prev_output_tokens):
        enc = self.encoder(src_tokens, src_lengths)
        return self.decoder(prev_output_tokens, encoder_out=enc)

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        """s2t_transformer.py:857-866 — fp32 (log-)softmax tagged batch_first."""
        if isinstance(net_output, (list, tuple)) and torch.is_tensor(net_output[0]) and net_output[0].dim() == 3:
            logits = net_output[0].float()
        else:
            logits = net_output.float()
        lp = torch.log_softmax(logits, -1) if log_probs else torch.softmax(logits, -1)
        lp.batch_first = True
        return lp

    def max_positions(self):
        return (self.encoder.max_positions(), self.decoder.max_positions())

    def set_num_updates(self, n):
        self.encoder.set_num_updates(n)


@register_model("s2t_ctc")
class S2TCTCModel(_HipModel):
    """models/speech_to_text/s2t_ctc.py:28-171 — encoder-only CTC model (``--encoder-type transformer | sate``)."""

    @classmethod
    def add_args(cls, parser):
        ref = reference_model_class("s2t_ctc")
        if ref is not None:
            ref.add_args(parser)

    def __init__(self, encoder):
        super().__init__()
        self.encoder = encoder
        self.flat = None

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)
        if getattr(args, "ctc_weight", 0) <= 0:
            args.ctc_weight = 1.0
        etype = getattr(args, "encoder_type", "transformer")
        if etype == "transformer":
            return cls(S2TTransformerEncoder(args, task))
        if etype == "sate":  # s2t_ctc.py:50-68: the embedding the textual encoder / XCTC head may share
            from .s2t_sate import S2TSATEEncoder, base_architecture as sate_base
            sate_base(args)
            tgt = task.target_dictionary
            embed = Embedding(len(tgt), args.encoder_embed_dim, tgt.pad())
            return cls(S2TSATEEncoder(args, task, embed))
        raise NotImplementedError("s2t_ctc --encoder-type %s" % etype)

    def forward(self, src_tokens, src_lengths, prev_output_tokens=None, **kwargs):
        return self.encoder(src_tokens, src_lengths, **kwargs)

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        logits = net_output["ctc_logit"][0] if isinstance(net_output, dict) else net_output[0]
        lp = torch.log_softmax(logits.float(), -1) if log_probs else torch.softmax(logits.float(), -1)
        lp.batch_first = False
        return lp


class CTCDecoder:
    """Greedy CTC decoding (models/speech_to_text/s2t_ctc.py:174-349): fp32 log-softmax -> arg-max (ties -> lowest
    id) -> padded frames to blank -> unique_consecutive -> drop blank, all on the GPU; one D2H copy of the ids."""

    def __init__(self, models, args=None, dictionary=None, blank_idx=0):
        self.model = models[0] if isinstance(models, (list, tuple)) else models
        self.blank = blank_idx
        self.pad = 1 if dictionary is None else dictionary.pad()
        # --ctc-inter-logit k (s2t_ctc.py:196,276-284): decode from the k-th intermediate head counted from the top
        self.ctc_inter_logit = int(getattr(args, "ctc_inter_logit", 0) or 0)
        if getattr(args, "ctc_self_ensemble", False):
            raise NotImplementedError("--ctc-self-ensemble (the reference's branch refers to an undefined name, s2t_ctc.py:316)")
        if int(getattr(args, "beam", 1) or 1) > 1 and getattr(args, "ctc_infer", "greedy") == "beam":
            raise NotImplementedError("CTC beam decoding (third-party ctcdecode in the reference)")
        # hypothesis["score"] of the reference sums the top-1 log-probability of every frame whose unmasked arg-max is not the
        # blank, PADDED frames included (s2t_ctc.py:327-329).  Packed rows (s2t_amd/rows.py) hold no padded frames: token ids are
        # the reference's, the score lacks the padded frames' (input-independent) term.  ``exact_scores`` = True decodes on the
        # padded layout, whose scores are the reference's (--ctc-exact-scores / S2T_CTC_EXACT_SCORES=1).
        import os
        self.exact_scores = bool(getattr(args, "ctc_exact_scores", False)) or os.environ.get("S2T_CTC_EXACT_SCORES", "0") == "1"
        self.fused_head = os.environ.get("S2T_CTC_HEAD_FUSED", "1") != "0"

    @torch.no_grad()
    def generate(self, models, sample, **kwargs):
        from . import kernels as K

        net_input = sample["net_input"]
        flagged = []
        if self.ctc_inter_logit != 0:  # the intermediate heads are decoded: they emit ctc_out_dtype (fp32 for bit-exact ids)
            for mod in self.model.modules():
                if hasattr(mod, "inter_ctc_layers") or hasattr(mod, "inter_xctc_layers"):
                    flagged.append((mod, getattr(mod, "decode_inter_logits", False)))
                    mod.decode_inter_logits = True
        was = Rows.ENABLED
        if self.exact_scores:
            Rows.ENABLED = False  # padded layout: the padded frames' logits exist and enter the score as in the reference
        # a plain Transformer / Conformer encoder whose final head is what gets decoded: head + arg-max in one launch, no logits
        # (S2T_CTC_HEAD_FUSED=0 keeps the two-kernel route)
        enc_mod = getattr(self.model, "encoder", None) or getattr(self.model, "e", None)
        fused = (self.fused_head and self.ctc_inter_logit == 0 and type(enc_mod) is S2TTransformerEncoder
                 and not getattr(enc_mod, "ctc_greedy_only", False))
        if fused:
            enc_mod.ctc_greedy_only = True
        try:
            enc = self.model(src_tokens=net_input["src_tokens"], src_lengths=net_input["src_lengths"])
        finally:
            Rows.ENABLED = was
            if fused:
                enc_mod.ctc_greedy_only = False
            # the flag is this decode's, not the model's: left set, later training / eval passes of the same model would refuse packed
            # rows and emit ctc_out_dtype logits from the intermediate heads
            for mod, old in flagged:
                mod.decode_inter_logits = old
        has_x = len(enc.get("xctc_logit", [])) > 0
        pk = enc.get("packed")
        if pk is not None and self.ctc_inter_logit == 0 and not has_x and pk.get("ctc_greedy") is not None:
            idx, top = pk["ctc_greedy"]
            return self._collapse_ids(idx, top, pk["rows"], pk["B"], pk["T"], rows=pk["rows"])
        if pk is not None and self.ctc_inter_logit == 0:  # packed rows (s2t_amd/rows.py): the decoded head's rows as they are
            l2d = pk.get("xctc_logit") if has_x else pk.get("ctc_logit")
            if l2d is not None:
                return self._collapse(l2d, pk["rows"], pk["B"], pk["T"], rows=pk["rows"])
        logit_tbv = enc["xctc_logit"][0] if has_x else enc["ctc_logit"][0]
        if isinstance(logit_tbv, (list, tuple)):
            logit_tbv = logit_tbv[0]
        mask = enc["encoder_padding_mask"][0]
        if self.ctc_inter_logit != 0:
            # s2t_ctc.py:262-284: the intermediate heads of the same family (XCTC when the encoder has one — note that the
            # reference then overwrites the list with the CTC family's whenever it is non-empty, :270-271)
            inter = enc.get("inter_xctc_logits", []) if has_x else []
            if not has_x or len(inter) > 0:
                inter = enc.get("inter_ctc_logits", [])
            if len(inter) != 0:
                assert self.ctc_inter_logit <= len(inter)
                item = inter[-self.ctc_inter_logit]
                if isinstance(item, (list, tuple)):
                    logit_tbv = item[0]
                    if len(item) >= 2 and item[1] is not None:
                        mask = item[1]
                # (a bare tensor entry leaves ctc_logit untouched in the reference, :278-283)
        Tn, B, V = logit_tbv.shape
        logits = logit_tbv.transpose(0, 1).reshape(B * Tn, V)  # batch-major rows (a view when it came from us)
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        lens = (~mask).sum(1).to(torch.int32)
        return self._collapse(logits, lens, B, Tn)

    def _collapse(self, logits, lens, B, Tn, rows=None):
        """Row arg-max and CTC collapse of batch-major logit rows ``[B * Tn, V]`` (``rows``: their packed geometry, if any —
        the frames behind an utterance's end are then not stored: they count as blank and add nothing to the score)."""
        from . import kernels as K

        V = logits.shape[1]
        dev = logits.device
        idx = torch.empty(B * Tn, dtype=torch.int32, device=dev)
        top = torch.empty(B * Tn, dtype=torch.float32, device=dev)
        K.argmax_lse(logits, logits.stride(0), B * Tn, V, idx, top, None, bound=rows)
        return self._collapse_ids(idx, top, lens, B, Tn, rows=rows)

    def _collapse_ids(self, idx, top, lens, B, Tn, rows=None):
        """CTC collapse of per-frame arg-max ids / log-probabilities (s2t_ctc.py:324-347)."""
        from . import kernels as K

        dev = idx.device
        toks = torch.zeros(B, Tn, dtype=torch.int64, device=dev)
        olen = torch.zeros(B, dtype=torch.int32, device=dev)
        osc = torch.zeros(B, dtype=torch.float32, device=dev)
        K.ctc_collapse(idx, top, lens, B, Tn, self.blank, toks, olen, osc, rows=rows)
        toks, olen, osc = toks.cpu(), olen.cpu(), osc.cpu()
        out = []
        for b in range(B):
            n = int(olen[b])
            hyp = toks[b, :n]
            out.append([{"tokens": hyp, "score": osc[b:b + 1], "attention": None, "alignment": None,
                         "positional_scores": torch.full((n,), float(osc[b]) / max(n, 1))}])
        return out


# ------------------------------------------------------------------------------------------------
# architectures (defaults as in s2t_transformer.py:2256-2470)
# ------------------------------------------------------------------------------------------------
def _d(args, name, val):
    if not hasattr(args, name):
        setattr(args, name, val)


@register_model_architecture("s2t_transformer", "s2t_transformer")
def base_architecture(args):
    _d(args, "input_feat_per_channel", 80)
    _d(args, "input_channels", 1)
    _d(args, "subsampling_type", "conv1d")
    _d(args, "subsampling_layers", 2)
    _d(args, "subsampling_filter", 1024)
    _d(args, "subsampling_kernel", 5)
    _d(args, "subsampling_stride", 2)
    _d(args, "subsampling_norm", "none")
    _d(args, "subsampling_activation", "glu")
    _d(args, "encoder_embed_dim", 512)
    _d(args, "encoder_ffn_embed_dim", 2048)
    _d(args, "encoder_layers", 12)
    _d(args, "encoder_attention_type", "selfattn")
    _d(args, "encoder_attention_heads", 8)
    _d(args, "encoder_normalize_before", True)
    _d(args, "decoder_embed_dim", args.encoder_embed_dim)
    _d(args, "decoder_ffn_embed_dim", args.encoder_ffn_embed_dim)
    _d(args, "decoder_layers", 6)
    _d(args, "decoder_attention_heads", 8)
    _d(args, "decoder_normalize_before", True)
    _d(args, "decoder_learned_pos", False)
    _d(args, "dropout", 0.1)
    _d(args, "attention_dropout", args.dropout)
    _d(args, "activation_dropout", args.dropout)
    _d(args, "activation_fn", "relu")
    _d(args, "share_decoder_input_output_embed", False)
    _d(args, "no_scale_embedding", False)
    _d(args, "encoder_no_scale_embedding", False)
    _d(args, "encoder_embed_norm", False)
    _d(args, "layer_padding_mask", False)
    _d(args, "ctc_layer", 0)
    _d(args, "ctc_weight", 0.0)
    _d(args, "share_ctc_and_embed", False)
    _d(args, "encoder_activation_fn", "relu")
    _d(args, "macaron_style", False)
    _d(args, "use_cnn_module", False)
    _d(args, "cnn_module_kernel", 31)
    _d(args, "cnn_module_norm", "batch_norm")
    _d(args, "max_source_positions", DEFAULT_MAX_SOURCE_POSITIONS)
    _d(args, "max_target_positions", DEFAULT_MAX_TARGET_POSITIONS)


@register_model_architecture("s2t_transformer", "s2t_transformer_s")
def s2t_transformer_s(args):
    _d(args, "encoder_embed_dim", 256)
    _d(args, "encoder_ffn_embed_dim", 256 * 8)
    _d(args, "encoder_attention_heads", 4)
    _d(args, "decoder_attention_heads", 4)
    _d(args, "dropout", 0.1)
    base_architecture(args)


@register_model_architecture("s2t_ctc", "s2t_ctc")
def s2t_ctc_base(args):
    base_architecture(args)


@register_model_architecture("s2t_ctc", "s2t_ctc_s")
def s2t_ctc_s(args):
    s2t_transformer_s(args)


def recipe_args(conformer=False, **over):
    """Namespace equal to the reference's YAML stack base.yaml + ctc.yaml (+ conformer.yaml)
    (egs/mustc/asr/conf/*.yaml); dropout defaults to 0 here (parity runs), pass dropout=0.1, attention_dropout=0.1,
    activation_dropout=0.1 for the recipe values."""
    a = Namespace(
        arch="s2t_transformer_s", share_decoder_input_output_embed=True, encoder_embed_norm=True,
        encoder_no_scale_embedding=True, subsampling_type="conv1d", subsampling_layers=2, subsampling_filter=1024,
        subsampling_kernel=5, subsampling_stride=2, subsampling_norm="none", subsampling_activation="glu",
        dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, activation_fn="relu", encoder_embed_dim=256,
        encoder_ffn_embed_dim=2048, encoder_layers=12, decoder_layers=6, encoder_attention_heads=4,
        decoder_embed_dim=256, decoder_ffn_embed_dim=2048, decoder_attention_heads=4, ctc_weight=0.3,
        share_ctc_and_embed=True, input_feat_per_channel=80, input_channels=1,
    )
    if conformer:
        a.macaron_style = True
        a.use_cnn_module = True
        a.cnn_module_kernel = 15
        a.encoder_attention_type = "rel_pos"
        a.encoder_activation_fn = "swish"
        a.layer_padding_mask = True
    for k, v in over.items():
        setattr(a, k, v)
    s2t_transformer_s(a)
    return a
