"""Autoregressive beam search on the HIP path.

Mirrors the contract of the reference's ``fairseq/sequence_generator.py`` (``SequenceGenerator.__init__`` :22-118,
``generate`` :176-189, ``_generate`` :191-614, ``finalize_hypos`` :650-786) with ``search.BeamSearch.step``
(``search.py:101-150``): same constructor arguments, same return value (per sentence a score-sorted list of dicts with
``tokens``, ``score``, ``attention``, ``alignment``, ``positional_scores``), same candidate rules —

  * step 0 expands the first beam only; later steps rank all (beam, token) pairs by cumulative log-probability;
  * pad and blank (``<s>``, index 0) are never selected, ``<unk>`` is penalised, ``</s>`` is forbidden before
    ``min_len`` and forced at ``max_len = min(int(a * src_len + b), max_decoder_positions - 1)``;
  * of the best ``2 * beam`` candidates an ``</s>`` among the first ``beam`` is finalised with
    ``score / (step + 1) ** len_penalty``; the first ``beam`` non-``</s>`` candidates continue;
  * a sentence stops once ``beam`` hypotheses are finished or ``step == max_len``.

The decoder runs incrementally (``TransformerDecoderScriptable`` with ``incremental_state``: per-layer key/value caches
in HBM, one query row per hypothesis, ``reorder_incremental_state`` after every step).  Finished sentences stay in the
batch (their rows are ignored) instead of being compacted away, which keeps every tensor shape static per step.
Joint CTC rescoring (``ctc_weight > 0``, reference :255-271 and :355-388): the reference pulls every hypothesis to the
host each step and scores it with ESPnet's numpy ``CTCPrefixScore`` through a dict keyed by the token string; here the
prefix states live in HBM ([rows, T', 2] fp32) and two launches of ``s2t_ctc_prefix_score`` per step score the
``int(1.5 * beam)`` candidates of every row and then advance the surviving rows' states — no per-token D2H.  The
reference scores all sentences against utterance 0 (``ctc_lprobs[0]``: a batch-size-1 path); this one scores each
sentence against its own frames, which is identical at batch size 1.
Not built (reference options that the recipes' generation configs leave off): sampling / diverse / constrained search,
LM fusion, n-gram blocking, prefix tokens.
"""
import math
from typing import Dict, List, Optional

import torch

from . import kernels as K

CTC_SCORING_RATIO = 1.5  # sequence_generator.py:19


class SequenceGenerator:
    def __init__(self, models, tgt_dict, beam_size=1, max_len_a=0, max_len_b=200, min_len=1, normalize_scores=True,
                 len_penalty=1.0, unk_penalty=0.0, temperature=1.0, match_source_len=False, no_repeat_ngram_size=0,
                 search_strategy=None, eos=None, symbols_to_strip_from_output=None, lm_model=None, lm_weight=1.0,
                 ctc_weight=0.0):
        models = list(models) if isinstance(models, (list, tuple)) else [models]
        if len(models) != 1:
            raise NotImplementedError("model ensembles")
        if match_source_len or no_repeat_ngram_size or search_strategy is not None or lm_model is not None:
            raise NotImplementedError("only plain beam search is built on the HIP path (see module docstring)")
        self.model = models[0]
        self.tgt_dict = tgt_dict
        self.pad, self.unk = tgt_dict.pad(), tgt_dict.unk()
        self.eos = tgt_dict.eos() if eos is None else eos
        self.blank = tgt_dict.bos()
        self.vocab_size = len(tgt_dict)
        self.beam_size = min(beam_size, self.vocab_size - 1)  # sequence_generator.py:74-75
        self.max_len_a, self.max_len_b, self.min_len = max_len_a, max_len_b, min_len
        self.normalize_scores, self.len_penalty, self.unk_penalty = normalize_scores, len_penalty, unk_penalty
        self.temperature = temperature
        self.ctc_weight = float(ctc_weight)
        assert temperature > 0, "--temperature must be greater than 0"

    @torch.no_grad()
    def generate(self, models, sample: Dict, **kwargs) -> List[List[Dict]]:
        return self._generate(sample)

    def _generate(self, sample):
        model = self.model
        model.eval()
        net_input = sample["net_input"]
        src_tokens, src_lengths = net_input["src_tokens"], net_input["src_lengths"]
        bsz, src_len = src_tokens.shape[:2]
        beam = self.beam_size
        dev = src_tokens.device
        max_len = min(int(self.max_len_a * src_len + self.max_len_b), model.decoder.max_positions() - 1)
        assert self.min_len <= max_len, "min_len cannot be larger than max_len, please adjust these!"

        enc = model.encoder(src_tokens, src_lengths)
        order = torch.arange(bsz, device=dev).view(-1, 1).repeat(1, beam).view(-1)
        ctc = None
        if self.ctc_weight > 0:
            ctc = self._ctc_prepare(enc, order, beam)
        enc = model.encoder.reorder_encoder_out(enc, order)

        tokens = torch.full((bsz * beam, max_len + 2), self.pad, dtype=torch.long, device=dev)
        tokens[:, 0] = self.eos
        scores = torch.zeros(bsz * beam, max_len + 1, dtype=torch.float32, device=dev)
        finalized: List[List[Dict]] = [[] for _ in range(bsz)]
        finished = [False] * bsz
        cand_size = 2 * beam
        bbsz_offsets = (torch.arange(bsz, device=dev) * beam).unsqueeze(1)
        cand_offsets = torch.arange(cand_size, device=dev)
        incremental_state: Dict = {}
        reorder: Optional[torch.Tensor] = None
        NEG = -math.inf

        for step in range(max_len + 1):
            if reorder is not None:
                model.decoder.reorder_incremental_state(incremental_state, reorder)
                enc = model.encoder.reorder_encoder_out(enc, reorder)
            logits, _ = model.decoder(tokens[:, :step + 1], encoder_out=enc, incremental_state=incremental_state)
            lprobs = torch.log_softmax(logits[:, -1, :].float() / self.temperature, dim=-1)
            if ctc is not None:
                self._ctc_rescore(ctc, lprobs, tokens, step)
            lprobs[lprobs != lprobs] = NEG
            lprobs[:, self.pad] = NEG
            lprobs[:, self.blank] = NEG
            lprobs[:, self.unk] -= self.unk_penalty
            if step >= max_len:
                lprobs[:, :self.eos] = NEG
                lprobs[:, self.eos + 1:] = NEG
            elif step < self.min_len:
                lprobs[:, self.eos] = NEG

            # ---- candidates (search.py:113-146)
            lp = lprobs.view(bsz, beam, self.vocab_size)
            if step == 0:
                lp = lp[:, :1, :]
            else:
                lp = lp + scores.view(bsz, beam, -1)[:, :, step - 1].unsqueeze(-1)
            flat = lp.reshape(bsz, -1)
            cand_scores, cand_flat = torch.topk(flat, k=min(cand_size, flat.size(1) - 1))
            cand_beams = torch.div(cand_flat, self.vocab_size, rounding_mode="trunc")
            cand_tokens = cand_flat.fmod(self.vocab_size)
            cand_bbsz = cand_beams + bbsz_offsets
            eos_mask = cand_tokens.eq(self.eos) & cand_scores.ne(NEG)

            # ---- finalise </s> candidates that sit in the first `beam` slots (host side: a handful of scalars)
            fin_mask = eos_mask[:, :beam].cpu()
            if bool(fin_mask.any()):
                cs, cb = cand_scores[:, :beam].cpu(), cand_bbsz[:, :beam].cpu()
                tok_cpu, sc_cpu = tokens[:, 1:step + 2].cpu(), scores[:, :step + 1].cpu()
                for sent in range(bsz):
                    if finished[sent]:
                        continue
                    for j in range(beam):
                        if fin_mask[sent, j] and len(finalized[sent]) < beam:
                            row = int(cb[sent, j])
                            hyp_tokens = tok_cpu[row].clone()
                            hyp_tokens[step] = self.eos
                            pos = sc_cpu[row].clone()
                            pos[step] = cs[sent, j]
                            pos[1:] = pos[1:] - pos[:-1].clone()
                            score = float(cs[sent, j])
                            if self.normalize_scores:
                                score /= (step + 1) ** self.len_penalty
                            finalized[sent].append({"tokens": hyp_tokens, "score": torch.tensor(score), "attention": None,
                                                    "alignment": torch.empty(0), "positional_scores": pos})
            for sent in range(bsz):
                if not finished[sent] and (len(finalized[sent]) == beam or step == max_len):
                    finished[sent] = True
            if all(finished) or step >= max_len:
                break

            # ---- the first `beam` candidates that are not </s> continue (sequence_generator.py:540-570)
            active_mask = eos_mask.to(cand_offsets.dtype) * cand_size + cand_offsets[:eos_mask.size(1)]
            _, active_hypos = torch.topk(active_mask, k=beam, dim=1, largest=False)
            active_bbsz = torch.gather(cand_bbsz, 1, active_hypos).view(-1)
            tokens[:, :step + 1] = tokens[:, :step + 1].index_select(0, active_bbsz)
            tokens.view(bsz, beam, -1)[:, :, step + 1] = torch.gather(cand_tokens, 1, active_hypos)
            if step > 0:
                scores[:, :step] = scores[:, :step].index_select(0, active_bbsz)
            scores.view(bsz, beam, -1)[:, :, step] = torch.gather(cand_scores, 1, active_hypos)
            reorder = active_bbsz
            if ctc is not None:
                self._ctc_advance(ctc, active_bbsz, tokens, step)

        for sent in range(bsz):
            finalized[sent].sort(key=lambda h: -float(h["score"]))
        return finalized

    # ---- joint CTC rescoring (reference :255-271, :355-388) -----------------------------------------------------------
    def _ctc_prepare(self, enc, order, beam):
        key = "xctc_logit" if len(enc.get("xctc_logit", [])) > 0 else "ctc_logit"  # :257-260
        logit = enc[key][0]  # (T', B, V), a view of the encoder's batch-major buffer
        Tn, B, V = logit.shape
        lp = torch.log_softmax(logit.transpose(0, 1).float(), dim=-1).reshape(B * Tn, V).contiguous()
        in_lens = (~enc["encoder_padding_mask"][0]).sum(1).to(torch.int32)
        sent = order.to(torch.int32).contiguous()
        R = sent.numel()
        state = torch.empty(R, Tn, 2, dtype=torch.float32, device=lp.device)
        K.ctc_prefix_init(lp, Tn, in_lens, sent, self.blank, state)
        return {"lp": lp, "T": Tn, "in_lens": in_lens, "sent": sent, "state": state,
                "prev": torch.zeros(R, dtype=torch.float32, device=lp.device),
                "row_len": in_lens.long().index_select(0, order), "k": min(V, int(beam * CTC_SCORING_RATIO))}

    def _ctc_rescore(self, ctc, lprobs, tokens, step):
        """lprobs[r, ids] <- (1 - w) * lprobs[r, ids] + w * (psi(prefix_r + ids) - psi(prefix_r)) for the ctc_beam best
        non-blank ids of every row, while step <= T' of the row's utterance (:355-382)."""
        masked = lprobs.clone()
        masked[:, self.blank] = -math.inf
        ids = torch.topk(masked, ctc["k"], dim=-1).indices.contiguous()
        psi = torch.empty(ids.shape, dtype=torch.float32, device=lprobs.device)
        K.ctc_prefix_score(ctc["lp"], ctc["T"], ctc["in_lens"], ctc["sent"], ctc["state"], tokens[:, step].contiguous(), step,
                           ids, self.blank, self.eos, psi)
        old = lprobs.gather(1, ids)
        new = (1.0 - self.ctc_weight) * old + self.ctc_weight * (psi - ctc["prev"].unsqueeze(1))
        live = (ctc["row_len"] >= step).unsqueeze(1)
        lprobs.scatter_(1, ids, torch.where(live, new, old))

    def _ctc_advance(self, ctc, parents, tokens, step):
        """The surviving rows' prefix states: parent state + the token just appended (tokens[:, step + 1]); ``tokens`` has
        already been re-ordered, so tokens[:, step] is the parent's last token."""
        parent_state = ctc["state"].index_select(0, parents)
        cand = tokens[:, step + 1].contiguous().view(-1, 1)
        psi = torch.empty(cand.shape, dtype=torch.float32, device=cand.device)
        new_state = torch.empty_like(parent_state).unsqueeze(1)
        K.ctc_prefix_score(ctc["lp"], ctc["T"], ctc["in_lens"], ctc["sent"], parent_state, tokens[:, step].contiguous(), step,
                           cand, self.blank, self.eos, psi, new_state)
        ctc["state"] = new_state.squeeze(1)
        ctc["prev"] = psi.view(-1)
