"""Joint label-smoothed cross-entropy + CTC criterion on the HIP path.

Reference: fairseq/criterions/label_smoothed_cross_entropy_with_ctc.py:24-237 and criterions/ctc.py:435-540
(``torch.nn.CTCLoss(blank=0, reduction="none", zero_infinity=True)`` on fp32 log-softmax, targets with pad/eos
stripped, summed over the batch).  ``forward(model, sample) -> (loss, sample_size, logging_output)``.
"""
import torch
import torch.nn as nn

from . import functional as Fn
from .registry import register_criterion


def ctc_targets(target, pad_idx, eos_idx):
    """criterions/ctc.py:516-540 — drop pad and eos; returns a left-packed (B, U) matrix and the label counts."""
    keep = (target != pad_idx) & (target != eos_idx)
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
    return target.gather(1, order).contiguous(), keep.sum(1).to(torch.int32)


@register_criterion("label_smoothed_cross_entropy_with_ctc")
class LabelSmoothedCrossEntropyCriterionWithCTC(nn.Module):
    def __init__(self, task, label_smoothing=0.1, sentence_avg=False, cfg=None, ctc_weight=0.0, inter_ctc_weight=None,
                 **unused):
        super().__init__()
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
        logits, _ = model.decoder(prev_output_tokens=ni["prev_output_tokens"], encoder_out=enc)
        target = sample["target"]
        B, U, V = logits.shape
        sums = Fn.label_smoothed_ce(logits.reshape(B * U, V), target.reshape(-1).contiguous(), self.eps, self.padding_idx)
        loss = sums[0]
        sample_size = target.size(0) if self.sentence_avg else sample["ntokens"]
        log = {"trans_loss": sums[0].detach(), "nll_loss": sums[1].detach(), "ntokens": sample["ntokens"],
               "nsentences": target.size(0), "sample_size": sample_size, "n_correct": sums[2].detach(),
               "total": sums[3].detach()}
        if self.ctc_weight > 0 and len(enc["ctc_logit"]) > 0:
            ctc_tbv = enc["ctc_logit"][0]
            Tn = ctc_tbv.shape[0]
            in_lens = (~enc["encoder_padding_mask"][0]).sum(1).to(torch.int32)
            tmat, tl = ctc_targets(target, self.padding_idx, self.eos_idx)
            l2d = ctc_tbv.transpose(0, 1).reshape(B * Tn, -1)  # a view: the encoder's buffer is batch-major
            ctc = Fn.ctc_loss(l2d, B, Tn, tmat, tl, in_lens, self.blank_idx)
            log["ctc_loss"] = ctc.detach()
            all_ctc = self.ctc_weight * ctc
            inter = enc.get("inter_ctc_logits", [])
            if self.inter_ctc_weight > 0 and len(inter) > 0:
                # criterions/ctc.py:568-633: every intermediate head against the same targets, averaged over the heads
                total = None
                for il in inter:
                    lg = il[0] if isinstance(il, (list, tuple)) else il
                    li = Fn.ctc_loss(lg.transpose(0, 1).reshape(B * Tn, -1), B, Tn, tmat, tl, in_lens, self.blank_idx)
                    total = li if total is None else total + li
                inter_loss = total / len(inter)
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
