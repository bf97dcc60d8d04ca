"""Joint label-smoothed cross-entropy + CTC criterion on the HIP path.

Reference: fairseq/criterions/label_smoothed_cross_entropy_with_ctc.py:24-237 and criterions/ctc.py:435-540
(``torch.nn.CTCLoss(blank=0, reduction="none", zero_infinity=True)`` on fp32 log-softmax, targets with pad/eos
stripped, summed over the batch).  ``forward(model, sample) -> (loss, sample_size, logging_output)``.
"""
import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .registry import criterion_base, register_criterion

_CriterionBase = criterion_base()


# S2T_CTC_SIDE=1 issues the CTC forward (alpha/beta: long thin launches) on a side stream beside the decoder.  Measured on
# the headline configuration it is 0.25 ms per step SLOWER than running it in line (21.6 vs 21.35 ms, A/B on one box), so
# it stays an opt-in experiment.
_CTC_SIDE = os.environ.get("S2T_CTC_SIDE", "0") == "1"


def ctc_targets(target, pad_idx, eos_idx):
    """criterions/ctc.py:516-540 — drop pad and eos; returns a left-packed (B, U) matrix and the label counts."""
    keep = (target != pad_idx) & (target != eos_idx)
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
    return target.gather(1, order).contiguous(), keep.sum(1).to(torch.int32)


def batch_bookkeeping(sample, pad_idx, eos_idx):
    """Everything about a batch that depends on its TARGETS only (CTC target matrix, label counts, the flattened target
    vector of the cross-entropy) — a dozen small sort / scan / gather launches.  Computed once per batch and kept
    (functional.batch_memo), like the collater's own fields (speech_to_text_dataset.py:411-485), instead of once per forward
    pass: a captured step then replays none of it."""
    def fn(target):
        tmat, tl = ctc_targets(target, pad_idx, eos_idx)
        return tmat, tl, target.reshape(-1).contiguous()

    def into(target, outs):  # the same into the memo's tensors, one launch (functional._recompute_in_place)
        K.ctc_targets(target, pad_idx, eos_idx, outs[0], outs[1])
        if outs[2].data_ptr() != target.data_ptr():
            outs[2].copy_(target.reshape(-1))

    if sample["target"].dtype == torch.int64 and sample["target"].is_contiguous():
        fn.into = into
    return Fn.batch_memo(("targets", pad_idx, eos_idx), (sample["target"],), fn)


@register_criterion("label_smoothed_cross_entropy_with_ctc")
class LabelSmoothedCrossEntropyCriterionWithCTC(_CriterionBase):
    def __init__(self, task, label_smoothing=0.1, sentence_avg=False, cfg=None, ctc_weight=0.0, inter_ctc_weight=None,
                 **unused):
        super().__init__(task)
        d = task.target_dictionary
        self.padding_idx, self.eos_idx = d.pad(), d.eos()
        self.blank_idx = 0
        self.eps = float(label_smoothing)
        self.sentence_avg = sentence_avg
        self.ctc_weight = ctc_weight
        # criterions/ctc.py:65,204 (CtcCriterionConfig.inter_ctc_weight; the reference reads it from ``cfg``)
        if inter_ctc_weight is None:
            inter_ctc_weight = float(getattr(cfg, "inter_ctc_weight", 0.0) or 0.0) if cfg is not None else 0.0
        self.inter_ctc_weight = float(inter_ctc_weight)
        self.report_accuracy = True

    def forward(self, model, sample, reduce=True, sync_logging=True):
        ni = sample["net_input"]
        enc = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
        target = sample["target"]
        B = target.size(0)
        # The CTC forward passes are issued before the decoder (optionally on a side stream, see _CTC_SIDE) and joined
        # before their values are used.  The reference computes them after the decoder
        # (label_smoothed_cross_entropy_with_ctc.py:95-128); the arithmetic is the same.
        ctc = inter_loss = None
        pk = enc.get("packed")
        if self.ctc_weight > 0 and pk is not None and pk.get("ctc_logit") is not None:
            # packed rows (s2t_amd/rows.py): the head's logits as the encoder left them, utterance b from row cu[b]
            tmat, tl, _ = batch_bookkeeping(sample, self.padding_idx, self.eos_idx)
            ctc = Fn.ctc_loss(pk["ctc_logit"], B, pk["T"], tmat, tl, pk["rows"], self.blank_idx, side=_CTC_SIDE, rows=pk["rows"])
            ipk = pk.get("inter_ctc_logit") or []
            if self.inter_ctc_weight > 0 and len(ipk) > 0:  # criterions/ctc.py:568-633 on the packed heads (same rows, same lengths)
                inter_loss = [Fn.ctc_loss(l2, B, pk["T"], tmat, tl, pk["rows"], self.blank_idx, side=_CTC_SIDE, rows=pk["rows"])
                              for l2 in ipk]
        elif self.ctc_weight > 0 and len(enc["ctc_logit"]) > 0:
            ctc_tbv = enc["ctc_logit"][0]
            Tn = ctc_tbv.shape[0]
            (in_lens,) = Fn.batch_memo(("ctc_in_lens", Fn.memo_owner(self)), (enc["encoder_padding_mask"][0],),
                                       lambda m: ((~m).sum(1).to(torch.int32),))
            tmat, tl, _ = batch_bookkeeping(sample, self.padding_idx, self.eos_idx)
            l2d = ctc_tbv.transpose(0, 1).reshape(B * Tn, -1)  # a view: the encoder's buffer is batch-major
            ctc = Fn.ctc_loss(l2d, B, Tn, tmat, tl, in_lens, self.blank_idx, side=_CTC_SIDE)
            inter = enc.get("inter_ctc_logits", [])
            if self.inter_ctc_weight > 0 and len(inter) > 0:
                # criterions/ctc.py:568-633: every intermediate head against the same targets, averaged over the heads
                inter_terms = []  # summed after the join: the values are being produced on the side stream
                for il in inter:
                    lg, il_lens = (il[0], il[1]) if isinstance(il, (list, tuple)) else (il, None)
                    # an entry carries the padding mask in force when it was produced (criterions/ctc.py:580-590): with
                    # CTC-guided compression the frame axis shrinks between the heads
                    il_lens = in_lens if il_lens is None else (~il_lens).sum(1).to(torch.int32)
                    Ti = lg.shape[0]
                    inter_terms.append(Fn.ctc_loss(lg.transpose(0, 1).reshape(B * Ti, -1), B, Ti, tmat, tl, il_lens,
                                                   self.blank_idx, side=_CTC_SIDE))
                inter_loss = inter_terms
        logits, dextra = model.decoder(prev_output_tokens=ni["prev_output_tokens"], encoder_out=enc, packed_out=True)
        tflat = batch_bookkeeping(sample, self.padding_idx, self.eos_idx)[2]
        dpk = dextra.get("packed") if isinstance(dextra, dict) else None
        if dpk is not None:
            # packed target rows (s2t_amd/rows.py): logits [B * U, V] with target b at rows cu[b] ...; the targets follow through
            # the row map (rows that hold no token: pad, which the loss skips)
            geom = K.rows_geom(dpk["rows"])
            pad_idx, U_, hdr = self.padding_idx, dpk["U"], geom.HEADER

            def pack_targets(t, buf):
                m = buf[hdr:]
                ok = m >= 0
                idx = torch.where(ok, (m >> 16).long() * U_ + (m & 0xffff).long(), torch.zeros_like(m, dtype=torch.long))
                return (torch.where(ok, t.reshape(-1)[idx], torch.full_like(idx, pad_idx)),)

            def pack_targets_into(t, buf, outs):
                K.gather_rows_i64(t, buf[hdr:], U_, pad_idx, outs[0])

            if tflat.dtype == torch.int64 and tflat.is_contiguous():
                pack_targets.into = pack_targets_into
            (tpk,) = Fn.batch_memo(("targets_packed", Fn.memo_owner(self)), (tflat, geom.buf), pack_targets)
            sums = Fn.label_smoothed_ce(logits, tpk, self.eps, self.padding_idx, rows=dpk["rows"])
        else:
            _, U, V = logits.shape
            sums = Fn.label_smoothed_ce(logits.reshape(B * U, V), tflat, self.eps, self.padding_idx)
        loss = sums[0]
        sample_size = target.size(0) if self.sentence_avg else sample["ntokens"]
        log = {"trans_loss": sums[0].detach(), "nll_loss": sums[1].detach(), "ntokens": sample["ntokens"],
               "nsentences": target.size(0), "sample_size": sample_size, "n_correct": sums[2].detach(),
               "total": sums[3].detach()}
        Fn.join_side_streams()
        if ctc is not None:
            log["ctc_loss"] = ctc.detach()
            all_ctc = self.ctc_weight * ctc
            if inter_loss is not None:
                inter_loss = sum(inter_loss[1:], inter_loss[0]) / len(enc["inter_ctc_logits"])
                log["inter_ctc_loss"] = inter_loss.detach()
                all_ctc = all_ctc + self.inter_ctc_weight * inter_loss
            log["all_ctc_loss"] = all_ctc.detach()
            loss = loss + all_ctc
        log["loss"] = loss.detach()
        if sync_logging:
            log = {k: (v.item() if torch.is_tensor(v) else v) for k, v in log.items()}
        return loss, sample_size, log

    @staticmethod
    def logging_outputs_can_be_summed():
        return True


@register_criterion("ctc")
class CtcCriterion(_CriterionBase):
    """criterions/ctc.py:156-1101 for encoder-only models (``s2t_ctc``): ``forward`` :258-281,
    ``get_ground_truth_alignment`` :283-433, ``compute_ctc_loss`` :542-1016.

        loss = ctc_weight * CTC(ctc_logit, transcript) + inter_ctc_weight * mean_i CTC(inter_ctc_logits[i], transcript)
             + xctc_weight * CTC(xctc_logit, target)   + inter_xctc_weight * mean_i CTC(inter_xctc_logits[i], target)

    every CTC summed over the batch with zero_infinity, fp32 log-softmax inside the loss kernel.  ``transcript`` =
    ``sample["transcript"]["tokens"]`` when present, else the target.  Logit entries may be the reference's lists
    ``[logit, padding_mask or None, force_emit]``; with ``ctc_masked_loss`` a force-emit entry switches the loss to the
    imputer loss (torch_imputer).  In training, when the encoder has a PAE ground-truth ratio, a first no-grad pass
    yields the alignment oracle that the second pass consumes (``ctc_alignment_oracle=``).
    Not built: AXCTC, self-distillation, entropy and mixup-consistency terms, the validation-time WER/CER counters."""

    def __init__(self, cfg=None, task=None, ctc_weight=1.0, save_dir=None, **over):
        super().__init__(task)
        d = task.target_dictionary
        self.pad_idx, self.eos_idx, self.blank_idx = d.pad(), d.eos(), 0

        def g(name, default=0.0):
            return over[name] if name in over else (getattr(cfg, name, default) if cfg is not None else default)

        self.sentence_avg = bool(g("sentence_avg", False))
        self.ctc_weight = float(ctc_weight)
        self.inter_ctc_weight = float(g("inter_ctc_weight"))
        self.xctc_weight = float(g("xctc_weight"))
        self.inter_xctc_weight = float(g("inter_xctc_weight"))
        self.ctc_masked_loss = bool(g("ctc_masked_loss", False))
        for name in ("axctc_weight", "inter_axctc_weight", "ctc_self_distill_weight", "xctc_self_distill_weight",
                     "ctc_entropy_weight", "ctc_mixup_consistent_weight", "inter_ctc_mixup_consistent_weight"):
            if float(g(name) or 0) != 0:
                raise NotImplementedError("CtcCriterion --%s on the HIP path" % name.replace("_", "-"))
        if g("inter_ctc_mlo", None) not in (None, ""):
            raise NotImplementedError("inter_ctc_mlo")
        self.use_ctc = self.ctc_weight + self.inter_ctc_weight > 0
        self.use_xctc = self.xctc_weight + self.inter_xctc_weight > 0

    # ---- criterions/ctc.py:283-433 ---------------------------------------------------------------------------------
    @torch.no_grad()
    def get_ground_truth_alignment(self, model, sample, **enc_kwargs):
        from .torch_imputer import best_alignment_states

        ni = sample["net_input"]
        enc = model.encoder(ni["src_tokens"], ni["src_lengths"], **enc_kwargs)
        mask = enc["ctc_padding_mask"][0] if "ctc_padding_mask" in enc else enc["encoder_padding_mask"][0]
        in_lens = (~mask).long().sum(-1)

        def pick(top, inter):
            lg = enc.get(top, [])
            lg = lg[0] if len(lg) else (enc.get(inter, [None])[-1] if len(enc.get(inter, [])) else None)
            return lg[0] if isinstance(lg, (list, tuple)) else lg

        def align(logit_tbv, tokens):
            lp = torch.log_softmax(logit_tbv.float(), dim=-1)  # (T, B, V)
            keep = (tokens != self.pad_idx) & (tokens != self.eos_idx)
            # (the reference pads the per-utterance lists with state 0 on the host, criterions/ctc.py:317-320; here the states
            # never leave the device)
            pad = best_alignment_states(lp, tokens, in_lens, keep.sum(-1), self.blank_idx, zero_infinity=True).to(tokens.dtype)
            pos = torch.div(pad, 2, rounding_mode="floor").clip(max=tokens.shape[1] - 1)
            oracle = tokens.gather(-1, pos)
            oracle.masked_fill_(pad % 2 == 0, self.blank_idx)
            mistake_flag = lp.argmax(dim=-1).transpose(0, 1) != oracle
            return oracle, pad, mistake_flag, mistake_flag.sum(-1) / in_lens

        out = {}
        ctc_logit = pick("ctc_logit", "inter_ctc_logits")
        if ctc_logit is not None:
            out["ctc"] = align(ctc_logit, sample["transcript"]["tokens"] if "transcript" in sample else sample["target"])
        xctc_logit = pick("xctc_logit", "inter_xctc_logits")
        if xctc_logit is not None:
            out["xctc"] = align(xctc_logit, self.get_ctc_target_text(sample))
        return out

    def get_ctc_target_text(self, sample):
        return sample["ctc_target"]["tokens"] if "ctc_target" in sample else sample["target"]

    # ---- one CTC term ----------------------------------------------------------------------------------------------
    def _ctc(self, entry, tmat, tl, in_lens):
        force_emit = None
        if isinstance(entry, (list, tuple)):
            logit = entry[0]
            if len(entry) > 1 and entry[1] is not None:
                in_lens = (~entry[1]).sum(-1).to(torch.int32)
            if len(entry) >= 3:
                force_emit = entry[2]
        else:
            logit = entry
        Tn, B, V = logit.shape
        if force_emit is not None and self.ctc_masked_loss:
            from .torch_imputer import imputer_loss
            lp = torch.log_softmax(logit.float(), dim=-1)
            return imputer_loss(lp, tmat, force_emit, in_lens, tl, blank=self.blank_idx, reduction="none",
                                zero_infinity=True).sum()
        l2d = logit.transpose(0, 1).reshape(B * Tn, V)  # a view: the encoders' buffers are batch-major
        return Fn.ctc_loss(l2d, B, Tn, tmat, tl, in_lens, self.blank_idx)

    def forward(self, model, sample, reduce=True, sync_logging=True, **enc_kwargs):
        ni = sample["net_input"]
        enc_kw = dict(enc_kwargs)
        first_kw = enc_kw.pop("first_pass_kwargs", {})
        if self.training and getattr(model.encoder, "pae_ground_truth_ratio", 0) != 0:
            enc_kw["ctc_alignment_oracle"] = self.get_ground_truth_alignment(model, sample, **first_kw)
        enc = model.encoder(ni["src_tokens"], ni["src_lengths"], **enc_kw)
        ntokens = sample["ntokens"]
        sample_size = sample["target"].size(0) if self.sentence_avg else ntokens
        log = {"ntokens": ntokens, "nsentences": sample["target"].size(0), "sample_size": sample_size}
        loss = self.compute_ctc_loss(model, sample, enc, log)
        log["loss"] = loss.detach()
        if sync_logging:
            log = {k: (v.item() if torch.is_tensor(v) else v) for k, v in log.items()}
        return loss, sample_size, log

    def compute_ctc_loss(self, model, sample, enc, log):
        transcript = sample["transcript"]["tokens"] if "transcript" in sample else sample["target"]
        mask = enc["ctc_padding_mask"][0] if "ctc_padding_mask" in enc else enc["encoder_padding_mask"][0]
        in_lens = (~mask).sum(-1).to(torch.int32)
        log["nfeatures"] = in_lens.sum()
        total = None

        def add(name, weight, value):
            nonlocal total
            log[name] = value.detach()
            total = weight * value if total is None else total + weight * value

        def mean_of(entries, tmat, tl):
            acc = None
            for e in entries:
                v = self._ctc(e, tmat, tl, in_lens)
                acc = v if acc is None else acc + v
            return acc / len(entries)

        if self.use_ctc:
            tmat, tl = ctc_targets(transcript, self.pad_idx, self.eos_idx)
            inter = enc.get("inter_ctc_logits", [])
            if self.inter_ctc_weight > 0 and len(inter) > 0:
                add("inter_ctc_loss", self.inter_ctc_weight, mean_of(inter, tmat, tl))
            if self.ctc_weight > 0 and len(enc.get("ctc_logit", [])) > 0:
                add("ctc_loss", self.ctc_weight, self._ctc(enc["ctc_logit"][0], tmat, tl, in_lens))
        if self.use_xctc:
            tmat, tl = ctc_targets(self.get_ctc_target_text(sample), self.pad_idx, self.eos_idx)
            inter = enc.get("inter_xctc_logits", [])
            if self.inter_xctc_weight > 0 and len(inter) > 0:
                add("inter_xctc_loss", self.inter_xctc_weight, mean_of(inter, tmat, tl))
            if self.xctc_weight > 0:
                assert len(enc.get("xctc_logit", [])) > 0
                add("xctc_loss", self.xctc_weight, self._ctc(enc["xctc_logit"][0], tmat, tl, in_lens))
        if total is None:
            raise RuntimeError("CtcCriterion: no CTC term is active for this model output")
        log["all_ctc_loss"] = total.detach()
        return total

    @staticmethod
    def logging_outputs_can_be_summed():
        return True
