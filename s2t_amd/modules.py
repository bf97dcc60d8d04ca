"""Host-side mirrors of the reference's modules on the S2T path (same class names, constructor arguments,
``state_dict`` keys and ``forward`` meaning), computing through the HIP kernels of ``libs2t_hip.so``.

Reference files (relative to /root/reference/fairseq): modules/speech_to_text/subsampling.py,
modules/s2t_transformer_layer.py, modules/multihead_attention.py, modules/espnet_multihead_attention.py,
modules/convolution.py, modules/layer_norm.py, modules/positional_encoding.py,
modules/sinusoidal_positional_embedding.py, modules/speech_to_text/ctc.py, modules/transformer_layer.py.

Internal activation layout is batch-major ``[B*T, C]`` (the reference is time-major ``(T, B, C)``); the
model classes transpose at the boundary.  Modules are parameter containers + a ``forward`` on 2-D row
matrices; there is no CPU fallback.
"""
import math

import torch
import torch.nn as nn

from . import functional as Fn


class Ctx:
    """Per-forward geometry shared by the layers of one encoder/decoder pass."""

    def __init__(self, B, T, lens_i32, dtype, pos_tab=None, mask_layers=False):
        self.B, self.T, self.lens, self.dtype = B, T, lens_i32, dtype
        self.pos_tab = pos_tab
        self.mask_layers = mask_layers
        # packed batch (s2t_amd/rows.py): the lengths tensor carries the geometry; launches without a mask of their own take it
        # as ``rows`` to stop at the live rows
        self.rows = lens_i32 if getattr(lens_i32, "_pk", None) is not None else None


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's key names
# ------------------------------------------------------------------------------------------------
class LayerNorm(nn.Module):
    """modules/layer_norm.py:30-35 (torch.nn.LayerNorm, eps 1e-5, affine)."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    def forward(self, x2d, lens=None, T=0, fork=False, rows=None):
        """``fork=True`` returns ``(LN(x), x)``: use the second value as the residual input of the block (see
        functional.LayerNormFn)."""
        return Fn.layer_norm(x2d, self.weight, self.bias, lens, T, fork, rows=rows)


class Linear(nn.Module):
    """nn.Linear parameter holder (xavier-uniform weight like fairseq's Linear helper)."""

    def __init__(self, in_f, out_f, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        nn.init.xavier_uniform_(self.weight)
        self.bias = nn.Parameter(torch.zeros(out_f)) if bias else None

    def forward(self, x2d, alpha=1.0, residual=None, out_dtype=None, rows=None):
        return Fn.linear(x2d, self.weight, self.bias, alpha, residual, out_dtype, rows=rows)


class _Conv1dK(nn.Module):
    """Conv1d weights stored [Cout][k][Cin] (GEMM layout); state_dict shows the reference's (Cout, Cin, k)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        w = torch.empty(cout, cin, k)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w.permute(0, 2, 1).contiguous())
        bound = 1 / math.sqrt(cin * k)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        self._register_state_dict_hook(self._to_ref)
        self._register_load_state_dict_pre_hook(self._from_ref)

    @staticmethod
    def _to_ref(module, sd, prefix, local_metadata):
        key = prefix + "weight"
        if key in sd:
            sd[key] = sd[key].permute(0, 2, 1).contiguous()

    def _from_ref(self, sd, prefix, *args):
        # state_dict() always shows (and load_state_dict always receives) the reference's (Cout, Cin, k) layout
        key = prefix + "weight"
        if key in sd and sd[key].dim() == 3:
            sd[key] = sd[key].permute(0, 2, 1).contiguous()


class Conv1dSubsampling(nn.Module):
    """modules/speech_to_text/subsampling.py:106-159 — 2 x [Conv1d(k5, s2, p2) -> GLU]; keys layers.{i}.0.*"""

    def __init__(self, num_layers, in_dim, filters, kernel_size, stride=2, norm="none", act="glu"):
        super().__init__()
        if num_layers != 2 or kernel_size != 5 or stride != 2 or norm != "none" or act != "glu":
            raise NotImplementedError("HIP subsampler covers the recipes' 2-layer k5/s2 conv1d + GLU configuration")
        chans = [in_dim, filters[0] // 2]
        outs = [filters[0], filters[1] * 2]
        self.layers = nn.ModuleList(
            [nn.ModuleList([_Conv1dK(chans[i], outs[i], kernel_size)]) for i in range(num_layers)]
        )

    @staticmethod
    def get_out_seq_lens_tensor(lens):
        for _ in range(2):
            lens = torch.div(lens - 1, 2, rounding_mode="floor") + 1
        return lens

    def forward(self, src_bt, out_lens_i32, dtype):
        c0, c1 = self.layers[0][0], self.layers[1][0]
        return Fn.subsample(src_bt, c0.weight, c0.bias, c1.weight, c1.bias, out_lens_i32, dtype)


class FeedForwardModule(nn.Module):
    """modules/s2t_transformer_layer.py:26-66 — keys w_1, w_2."""

    def __init__(self, input_feat, hidden_units, dropout1, dropout2, activation_fn="relu", bias=True):
        super().__init__()
        self.w_1 = Linear(input_feat, hidden_units)
        self.w_2 = Linear(hidden_units, input_feat)
        self.activation_fn = activation_fn
        self.dropout1, self.dropout2 = float(dropout1 or 0.0), float(dropout2 or 0.0)

    def forward(self, x_ln, residual, scale):
        return Fn.ffn(x_ln, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias, self.activation_fn, scale,
                      residual, self.dropout1, self.dropout2, self.training)

    def block(self, x, norm, scale, end_norm=None, end_lens=None, end_T=0, rows=None):
        """x + scale * ffn(norm(x)) [-> end_norm]: the whole pre-LN block, one launch when the row-block kernel applies."""
        en = (end_norm.weight, end_norm.bias) if end_norm is not None else None
        return Fn.ffn_block(x, norm.weight, norm.bias, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias,
                            self.activation_fn, scale, self.dropout1, self.dropout2, self.training, en, end_lens, end_T,
                            rows=rows)


class MultiheadAttention(nn.Module):
    """modules/multihead_attention.py:24-431 — keys k_proj, v_proj, q_proj, out_proj."""

    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0, bias=True, self_attention=False,
                 encoder_decoder_attention=False, **unused):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.self_attention, self.encoder_decoder_attention = self_attention, encoder_decoder_attention
        self.attn_dropout = float(dropout or 0.0)  # on the probabilities (multihead_attention.py:411-413)
        self.out_dropout = 0.0                     # the owning layer's dropout_module on the branch output
        self.k_proj = Linear(embed_dim, embed_dim)
        self.v_proj = Linear(embed_dim, embed_dim)
        self.q_proj = Linear(embed_dim, embed_dim)
        self.out_proj = Linear(embed_dim, embed_dim)
        for p in (self.k_proj, self.v_proj, self.q_proj):  # multihead_attention.py:96-104
            nn.init.xavier_uniform_(p.weight, gain=1 / math.sqrt(2))

    def flat_groups(self):
        if self.self_attention:
            return [[self.q_proj.weight, self.k_proj.weight, self.v_proj.weight],
                    [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]]
        return [[self.k_proj.weight, self.v_proj.weight], [self.k_proj.bias, self.v_proj.bias]]

    def _prm(self):
        return {"q_w": self.q_proj.weight, "q_b": self.q_proj.bias, "k_w": self.k_proj.weight, "k_b": self.k_proj.bias,
                "v_w": self.v_proj.weight, "v_b": self.v_proj.bias, "o_w": self.out_proj.weight, "o_b": self.out_proj.bias}

    def forward(self, xq, xkv, residual, B, Tq, Tk, key_lens=None, causal=False, norm=None, kv=None, q_rows=None, chain=None):
        """``norm``: the LayerNorm in front of a self-attention block — ``xq`` is then the un-normalised block input and
        the residual (pass ``residual=None``).  ``kv``: keys / values projected for the whole stack (Fn.cross_kv).
        ``q_rows``: the packed geometry of the QUERY rows of an encoder-decoder attention (key side: ``key_lens``)."""
        return Fn.attention(xq, xkv, residual, self._prm(), self.num_heads, B, Tq, Tk, key_lens, causal, "abs", None,
                            self.attn_dropout, self.out_dropout, self.training,
                            ln=(norm.weight, norm.bias) if norm is not None else None, kv=kv, q_rows=q_rows, chain=chain)


class RelPositionMultiHeadedAttention(nn.Module):
    """modules/espnet_multihead_attention.py:270-356 — keys linear_q/k/v/out, linear_pos, pos_bias_u/v."""

    def __init__(self, n_feat, n_head, dropout, zero_triu=False):
        super().__init__()
        self.h, self.d_k = n_head, n_feat // n_head
        self.attn_dropout = float(dropout or 0.0)  # espnet_multihead_attention.py:41,147
        self.out_dropout = 0.0
        if zero_triu:
            raise NotImplementedError("zero_triu")
        self.linear_q = Linear(n_feat, n_feat)
        self.linear_k = Linear(n_feat, n_feat)
        self.linear_v = Linear(n_feat, n_feat)
        self.linear_out = Linear(n_feat, n_feat)
        self.linear_pos = Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.empty(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.empty(self.h, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)

    def flat_groups(self):
        return [[self.linear_q.weight, self.linear_k.weight, self.linear_v.weight],
                [self.linear_q.bias, self.linear_k.bias, self.linear_v.bias]]

    def _prm(self):
        return {"q_w": self.linear_q.weight, "q_b": self.linear_q.bias, "k_w": self.linear_k.weight,
                "k_b": self.linear_k.bias, "v_w": self.linear_v.weight, "v_b": self.linear_v.bias,
                "o_w": self.linear_out.weight, "o_b": self.linear_out.bias, "pos_w": self.linear_pos.weight,
                "pos_u": self.pos_bias_u, "pos_v": self.pos_bias_v}

    def forward(self, x, residual, B, T, key_lens, pos_tab, norm=None, pos_p=None, chain=None):
        return Fn.attention(x, None, residual, self._prm(), self.h, B, T, T, key_lens, False, "rel", pos_tab,
                            self.attn_dropout, self.out_dropout, self.training,
                            ln=(norm.weight, norm.bias) if norm is not None else None, pos_p=pos_p, chain=chain)


class _BatchNorm1d(nn.Module):
    """nn.BatchNorm1d holder (keys weight, bias, running_mean, running_var, num_batches_tracked)."""

    def __init__(self, c, momentum=0.1):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.momentum = momentum


class _ConvW(nn.Module):
    def __init__(self, shape, fan_in):
        super().__init__()
        bound = 1 / math.sqrt(fan_in)
        self.weight = nn.Parameter(torch.empty(*shape).uniform_(-bound, bound))


class ConvolutionModule(nn.Module):
    """modules/convolution.py:11-120 — keys pointwise_conv1, depthwise_conv, norm, pointwise_conv2 (no biases)."""

    def __init__(self, embed_dim, expand_embed_dim, depthwise_kernel_size, dropout, activation_fn="swish", bias=False,
                 stride=1, padding=None, export=False, norm_type="batch_norm"):
        super().__init__()
        if bias or stride != 1 or padding is not None or norm_type != "batch_norm" or embed_dim != expand_embed_dim:
            raise NotImplementedError("HIP conv module covers the recipes' configuration (batch_norm, no bias, stride 1)")
        self.dropout_p = float(dropout or 0.0)
        d, k = embed_dim, depthwise_kernel_size
        self.pointwise_conv1 = _ConvW((2 * d, d, 1), d)
        self.depthwise_conv = _ConvW((d, 1, k), k)
        self.norm = _BatchNorm1d(d)
        self.pointwise_conv2 = _ConvW((d, d, 1), d)
        self.activation_fn = activation_fn

    def forward(self, x_ln_masked, residual, B, T, lens, norm=None):
        """``norm``: conv_norm — the first argument is then the un-normalised block input and the residual."""
        prm = {"pw1_w": self.pointwise_conv1.weight, "dw_w": self.depthwise_conv.weight, "bn_w": self.norm.weight,
               "bn_b": self.norm.bias, "pw2_w": self.pointwise_conv2.weight}
        buf = {"running_mean": self.norm.running_mean, "running_var": self.norm.running_var}
        if self.training and not getattr(self, "counter_elsewhere", False):
            self.norm.num_batches_tracked += 1  # (the encoders bump all their layers' counters with one launch)
        return Fn.conv_module(x_ln_masked, residual, prm, buf, self.activation_fn, B, T, lens, self.training,
                              self.norm.momentum, self.dropout_p, ln=(norm.weight, norm.bias) if norm is not None else None)


class _ConvWB(nn.Module):
    """Conv1d weight + bias holder with nn.Conv1d's default initialisation."""

    def __init__(self, shape, fan_in):
        super().__init__()
        bound = 1 / math.sqrt(fan_in)
        self.weight = nn.Parameter(torch.empty(*shape).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(shape[0]).uniform_(-bound, bound))


class DownSampleConvolutionModule(nn.Module):
    """fairseq/modules/downsample_convolution.py:15-123 — keys pointwise_conv1, depthwise_conv, norm, pointwise_conv2 (all
    with biases): mask -> pointwise conv -> depthwise conv with kernel = stride = ratio, no padding -> BatchNorm ->
    Swish -> pointwise conv -> mask at floor(len / ratio).  Used by the PDS multi-scale fusion
    (pdss2t_transformer.py:1187-1233, transform ``conv2``)."""

    def __init__(self, channels, kernel_size, input_channels=None, stride=1):
        super().__init__()
        if kernel_size != stride:
            raise NotImplementedError("DownSampleConvolutionModule with kernel != stride")
        cin = input_channels or channels
        self.stride = stride
        self.pointwise_conv1 = _ConvWB((channels, cin, 1), cin)
        self.depthwise_conv = _ConvWB((channels, 1, kernel_size), kernel_size)
        self.norm = _BatchNorm1d(channels)
        self.pointwise_conv2 = _ConvWB((channels, channels, 1), channels)

    def forward(self, x, B, T, lens):
        """x [B*T, Cin] -> ([B*(T // stride), C], T // stride, new lens)."""
        x = MaskRows.apply(x, lens, T)
        y = Fn.linear(x, self.pointwise_conv1.weight, self.pointwise_conv1.bias)
        prm = {"dw_w": self.depthwise_conv.weight, "dw_b": self.depthwise_conv.bias, "bn_w": self.norm.weight,
               "bn_b": self.norm.bias}
        buf = {"running_mean": self.norm.running_mean, "running_var": self.norm.running_var}
        if self.training:
            self.norm.num_batches_tracked += 1
        a = Fn.pool_bn_act(y, prm, buf, "swish", B, T, self.stride, self.training, self.norm.momentum)
        To = T // self.stride
        y = Fn.linear(a, self.pointwise_conv2.weight, self.pointwise_conv2.bias)
        out_lens = torch.div(lens, self.stride, rounding_mode="floor").to(torch.int32)
        return MaskRows.apply(y, out_lens, To), To, out_lens


class S2TTransformerEncoderLayer(nn.Module):
    """modules/s2t_transformer_layer.py:69-322 (pre-LN; macaron / conv-module / rel_pos variants)."""

    def __init__(self, args, embed_dim=None, ffn_dim=None, num_heads=None, conv_activation=None, cnn_kernel=None):
        super().__init__()
        d = embed_dim or args.encoder_embed_dim
        ffn_dim = ffn_dim or args.encoder_ffn_embed_dim
        heads = num_heads or args.encoder_attention_heads
        if not args.encoder_normalize_before:
            raise NotImplementedError("post-LN encoder layers")
        self.attn_type = getattr(args, "encoder_attention_type", "selfattn")
        if self.attn_type == "selfattn":
            self.self_attn = MultiheadAttention(d, heads, dropout=args.dropout, self_attention=True)
        elif self.attn_type == "rel_pos":
            self.self_attn = RelPositionMultiHeadedAttention(d, heads, dropout=args.dropout)
        else:
            raise NotImplementedError("encoder attention type %s (HIP path: selfattn, rel_pos)" % self.attn_type)
        self.self_attn.out_dropout = float(args.dropout or 0.0)  # dropout_module(x) before the residual (:291)
        self.self_attn_layer_norm = LayerNorm(d)
        act = getattr(args, "encoder_activation_fn", "relu")
        if args.macaron_style:
            self.macaron_ffn = FeedForwardModule(d, ffn_dim, args.dropout, args.dropout, act)
            self.macaron_norm = LayerNorm(d)
            self.ffn_scale = 0.5
        else:
            self.macaron_ffn = self.macaron_norm = None
            self.ffn_scale = 1.0
        if args.use_cnn_module:
            self.conv_norm = LayerNorm(d)
            # NB: the conv-module activation is --activation-fn, not --encoder-activation-fn (s2t_transformer_layer.py:125)
            self.conv_module = ConvolutionModule(d, d, depthwise_kernel_size=cnn_kernel or args.cnn_module_kernel,
                                                 dropout=args.dropout,
                                                 activation_fn=conv_activation or getattr(args, "activation_fn", "swish"),
                                                 norm_type=getattr(args, "cnn_module_norm", "batch_norm"))
            self.final_norm = LayerNorm(d)
        else:
            self.conv_norm = self.conv_module = self.final_norm = None
        self.ffn = FeedForwardModule(d, ffn_dim, args.dropout, args.dropout, act)
        self.ffn_norm = LayerNorm(d)

    def forward(self, x, c: Ctx, mask_output: bool):
        """x: [B*T, d].  ``mask_output``: zero padded frames of the result (the NEXT layer's layer_padding_mask)."""
        B, T, lens = c.B, c.T, c.lens
        if self.macaron_norm is not None:
            x = self.macaron_ffn.block(x, self.macaron_norm, self.ffn_scale, rows=c.rows)
        # the attention output projection and the convolution module's conv_norm + pointwise conv 1 + GLU are row-local
        # neighbours: one launch where the row-block kernels apply (functional.attention, s2t_rowblock_chain)
        chain = None
        if self.conv_module is not None:
            d = x.shape[1]
            chain = {"w1": Fn.cw(self.conv_module.pointwise_conv1.weight).view(2 * d, d), "ln_g": self.conv_norm.weight,
                     "ln_b": self.conv_norm.bias, "lens": lens, "T": T}
        if self.attn_type == "rel_pos":
            x = self.self_attn(x, None, B, T, lens, c.pos_tab, norm=self.self_attn_layer_norm,
                               pos_p=getattr(c, "cur_pos_p", None), chain=chain)
        else:
            x = self.self_attn(x, None, None, B, T, T, lens, norm=self.self_attn_layer_norm, chain=chain)
        if self.conv_module is not None:
            x = self.conv_module(x, None, B, T, lens, norm=self.conv_norm)  # conv input mask fused (convolution.py:86-88)
        x = self.ffn.block(x, self.ffn_norm, self.ffn_scale, self.final_norm, lens if mask_output else None, T, rows=c.rows)
        if self.final_norm is None and mask_output:
            x = MaskRows.apply(x, lens, T)
        return x


class MaskRows(torch.autograd.Function):
    """x.masked_fill(pad, 0) on a row matrix (s2t_transformer.py:1828-1836)."""

    @staticmethod
    def forward(ctx, x, lens, T):
        from . import kernels as K

        y = x.clone()
        K.mask_rows(y, lens, y.shape[0], T, y.shape[1])
        ctx.lens, ctx.T = lens, T
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import kernels as K

        g = dy.clone()
        K.mask_rows(g, ctx.lens, g.shape[0], ctx.T, g.shape[1])
        return g, None, None


class CTC(nn.Module):
    """modules/speech_to_text/ctc.py:17-75 — Linear(d -> V, bias), init N(0, d^-0.5)."""

    def __init__(self, embed_dim, dictionary_size, dropout, need_layernorm=False, dictionary=None):
        super().__init__()
        self.dropout_p = float(dropout or 0.0)
        self.ctc_projection = Linear(embed_dim, dictionary_size)
        nn.init.normal_(self.ctc_projection.weight, mean=0, std=embed_dim ** -0.5)
        nn.init.constant_(self.ctc_projection.bias, 0.0)
        self.LayerNorm = LayerNorm(embed_dim) if need_layernorm else None
        self.dictionary = dictionary
        self.infer_decoding = False
        self.blank_idx = 0

    def set_infer(self, is_infer, text_post_process, dictionary, path):
        self.infer_decoding, self.dictionary = is_infer, dictionary

    def forward(self, x2d, out_dtype=None, rows=None):
        if self.LayerNorm is not None:
            x2d = self.LayerNorm(x2d, rows=rows)
        x2d = Fn.dropout(x2d, self.dropout_p, self.training)  # ctc_dropout_module (ctc.py:59)
        return self.ctc_projection(x2d, out_dtype=out_dtype, rows=rows)

    def greedy_supported(self, x2d):
        """The head + arg-max in one launch (csrc/ctc_head.hip) applies: inference, bf16 rows of width 256."""
        from . import kernels as K

        return (not self.training and not torch.is_grad_enabled() and x2d.dtype == torch.bfloat16
                and K.ctc_head_greedy_supported(x2d, Fn.cw(self.ctc_projection.weight)))

    def greedy(self, x2d, rows=None):
        """-> (idx int32 [M], top_lp fp32 [M]): per row the first arg-max of the head's fp32 logits and its log-probability
        (s2t_ctc.py:312-328), the [M, V] logits never stored: what ``forward(out_dtype=float32)`` followed by ``s2t_argmax_lse`` gives,
        up to the summation order of the fp32 products inside a logit."""
        from . import kernels as K

        if self.LayerNorm is not None:
            x2d = self.LayerNorm(x2d, rows=rows)
        M = x2d.shape[0]
        idx = torch.zeros(M, dtype=torch.int32, device=x2d.device)   # (rows beyond a packed batch's live ones stay blank / 0)
        top = torch.zeros(M, dtype=torch.float32, device=x2d.device)
        b = self.ctc_projection.bias
        K.ctc_head_greedy(x2d, Fn.cw(self.ctc_projection.weight), b.data if b is not None else None, idx, top, None, bound=rows)
        return idx, top


class Adapter(nn.Module):
    """modules/speech_to_text/adapter.py:89-349 — `inter_league`: x + dist @ embed_adapter.weight with
    dist = softmax(ctc_logit / tau); with ``gt_ratio`` > 0 and an oracle, the flagged frames take the one-hot
    (``oracle_smooth``: 0.9 / 0.1-spread) distribution of the ground-truth label (:245-262)."""

    def __init__(self, dim, adapter_type, dictionary_size, embed_tokens=None, strategy=None):
        super().__init__()
        if adapter_type not in ("inter_league", "none"):
            raise NotImplementedError("adapter %s (HIP path: inter_league, none)" % adapter_type)
        strategy = strategy or {}
        for k in ("embed_norm", "out_norm", "gumbel", "distribution_hard"):
            if strategy.get(k):
                raise NotImplementedError("adapter strategy %s" % k)
        self.adapter_type = adapter_type
        if adapter_type == "inter_league":
            if embed_tokens is not None:
                self.embed_adapter = embed_tokens
            else:  # nn.Linear(dim, V, bias=False) in the reference ("reverse for initialization"), N(0, dim^-0.5)
                self.embed_adapter = Linear(dim, dictionary_size, bias=False)
                nn.init.normal_(self.embed_adapter.weight, mean=0, std=dim ** -0.5)
        self.temperature = float(strategy.get("ctc_temperature", strategy.get("distribution_temperature", 1.0)))
        self.ground_truth_ratio = float(strategy.get("gt_ratio", 0) or 0)
        self.oracle_smooth = bool(strategy.get("oracle_smooth", False))

    def forward(self, x2d, logit2d, oracle=None, oracle_mask=None, rows=None):
        """x2d [B*T, d], logit2d [B*T, V]; oracle / oracle_mask (B, T) as produced by ``pae_oracle_mask``; ``rows``: the
        lengths tensor of a packed batch whose rows x2d / logit2d hold (s2t_amd/rows.py)."""
        if self.adapter_type == "none":
            return x2d
        if self.ground_truth_ratio > 0 and oracle is not None:
            return Fn.adapter_inter_league(x2d, logit2d, self.embed_adapter.weight, self.temperature,
                                           oracle.reshape(-1), oracle_mask.reshape(-1), self.oracle_smooth)
        return Fn.adapter_inter_league(x2d, logit2d, self.embed_adapter.weight, self.temperature, rows=rows)


def pae_oracle_mask(entry, gt_ratio, adaptive=False, only_mistake=False, mask=None):
    """s2t_transformer.py:1904-1935 / s2t_sate.py:774-797: frames that are fed the ground-truth label.
    entry = (oracle, best_aligns_pad, mistake_flag, mistake_ratio) from CtcCriterion.get_ground_truth_alignment;
    ``mask`` (B, T) bool replaces the ``torch.rand(...) < prob`` draw (tests replay the reference's draw).
    -> (oracle, mask, force_emit)"""
    oracle, aligns, mistake_flag, mistake_ratio = entry
    if mask is None:
        prob = gt_ratio * mistake_ratio.unsqueeze(-1) if adaptive else gt_ratio
        mask = torch.rand(oracle.size(), device=oracle.device) < prob
    mask = mask.to(device=oracle.device, dtype=torch.bool).clone()
    if only_mistake:
        mask &= mistake_flag
    return oracle, mask, aligns.masked_fill(~mask, -1)


class TransformerDecoderLayer(nn.Module):
    """modules/transformer_layer.py:240-546 (pre-LN) — keys self_attn, encoder_attn, *_layer_norm, fc1, fc2."""

    def __init__(self, args):
        super().__init__()
        d = args.decoder_embed_dim
        if not args.decoder_normalize_before:
            raise NotImplementedError("post-LN decoder layers")
        self.self_attn = MultiheadAttention(d, args.decoder_attention_heads, dropout=args.attention_dropout,
                                            self_attention=True)
        self.self_attn_layer_norm = LayerNorm(d)
        self.encoder_attn = MultiheadAttention(d, args.decoder_attention_heads, dropout=args.attention_dropout,
                                               encoder_decoder_attention=True)
        self.encoder_attn_layer_norm = LayerNorm(d)
        self.fc1 = Linear(d, args.decoder_ffn_embed_dim)
        self.fc2 = Linear(args.decoder_ffn_embed_dim, d)
        self.final_layer_norm = LayerNorm(d)
        self.activation_fn = getattr(args, "activation_fn", "relu")
        # transformer_layer.py:271-280: dropout_module on every branch output, activation_dropout after fc1's activation
        self.dropout_p = float(args.dropout or 0.0)
        self.activation_dropout_p = float(getattr(args, "activation_dropout", 0) or 0.0)
        self.self_attn.out_dropout = self.dropout_p
        self.encoder_attn.out_dropout = self.dropout_p

    def step(self, x, state, n_keys, mem, Bb, Tm, mem_lens):
        """One incremental decoding step (inference): ``x`` [Bb, d] is the current position; ``state`` holds this
        layer's caches: "self_kv" [Bb, cap, 2d] (grown geometrically) and "mem_kv" [Bb, Tm, 2d] (projected once)."""
        d = x.shape[1]
        kv = state.get("self_kv")
        if kv is None or kv.shape[1] < n_keys:
            cap = max(16, 2 * (n_keys - 1), n_keys)
            new = torch.zeros(Bb, cap, 2 * d, dtype=x.dtype, device=x.device)
            if kv is not None:
                new[:, :kv.shape[1]].copy_(kv)
            state["self_kv"] = kv = new
        if "mem_kv" not in state:
            state["mem_kv"] = Fn.project_memory(mem, self.encoder_attn._prm(), Bb * Tm).view(Bb, Tm, 2 * d)
        y = self.self_attn_layer_norm(x)
        x = Fn.attention_step(y, x, self.self_attn._prm(), self.self_attn.num_heads, kv, n_keys, None, True)
        y = self.encoder_attn_layer_norm(x)
        x = Fn.attention_step(y, x, self.encoder_attn._prm(), self.encoder_attn.num_heads, state["mem_kv"], Tm, mem_lens,
                              False)
        y = self.final_layer_norm(x)
        return Fn.ffn(y, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.activation_fn, 1.0, x,
                      0.0, 0.0, False)

    def forward(self, x, mem, B, U, Tm, self_lens, mem_lens, mem_kv=None):
        # (the LayerNorm in front of the self-attention rides in the fused q|k|v projection's prologue where the row-block
        # kernel applies, its backward in that projection's input-gradient kernel)
        # packed target rows (s2t_amd/rows.py): ``self_lens`` then carries their geometry — the key side of the self-attention,
        # the QUERY side of the encoder-decoder attention, the live-row bound of everything without a mask
        rows = self_lens if getattr(self_lens, "_pk", None) is not None else None
        x = self.self_attn(x, None, None, B, U, U, self_lens, causal=True, norm=self.self_attn_layer_norm)
        if mem_kv is not None:  # (likewise the LayerNorm in front of the encoder-decoder attention, in its query projection)
            x = self.encoder_attn(x, None, None, B, U, Tm, mem_lens, kv=mem_kv, norm=self.encoder_attn_layer_norm, q_rows=rows)
        else:
            y, x = self.encoder_attn_layer_norm(x, fork=True, rows=rows)
            x = self.encoder_attn(y, mem, x, B, U, Tm, mem_lens, q_rows=rows)
        # (one launch for the whole block where the row-block kernel applies: at the decoder's few thousand rows the hidden units
        # of a 128-row block are dealt to eight workgroups, csrc/ffn_pc.hip)
        return Fn.ffn_block(x, self.final_layer_norm.weight, self.final_layer_norm.bias, self.fc1.weight, self.fc1.bias,
                            self.fc2.weight, self.fc2.bias, self.activation_fn, 1.0, self.activation_dropout_p, self.dropout_p,
                            self.training, rows=rows)


# ------------------------------------------------------------------------------------------------
# positional tables (host-built once, cached on device)
# ------------------------------------------------------------------------------------------------
def sinusoidal_table(n_pos: int, dim: int, padding_idx: int = 1) -> torch.Tensor:
    """modules/sinusoidal_positional_embedding.py:36-58: [sin | cos] halves, step log(1e4)/(dim/2-1), pad row zero."""
    half = dim // 2
    step = math.log(10000.0) / (half - 1)
    inv = torch.exp(torch.arange(half, dtype=torch.float32) * -step)
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(n_pos, 1)], dim=1)
    tab[padding_idx] = 0.0
    return tab


def rel_pos_table(T: int, dim: int) -> torch.Tensor:
    """modules/positional_encoding.py:121-166: (2T-1, dim), row n <-> offset T-1-n, interleaved sin/cos."""
    rel = torch.arange(T - 1, -T, -1, dtype=torch.float32)[:, None]
    inv = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    tab = torch.zeros(2 * T - 1, dim)
    tab[:, 0::2] = torch.sin(rel * inv)
    tab[:, 1::2] = torch.cos(rel * inv)
    return tab


class _TableCache:
    def __init__(self):
        self.c = {}

    def get(self, kind, n, dim, device, dtype=torch.float32):
        key = (kind, n, dim, str(device), dtype)
        if key not in self.c:
            t = sinusoidal_table(n, dim) if kind == "sin" else rel_pos_table(n, dim)
            self.c[key] = t.to(device=device, dtype=dtype)
        return self.c[key]


TABLES = _TableCache()
