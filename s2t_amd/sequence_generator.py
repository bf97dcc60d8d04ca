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
Not built (reference options that the recipes' generation configs leave off): sampling / diverse / constrained search,
LM fusion, n-gram blocking, prefix tokens, joint CTC prefix rescoring (the reference scores prefixes on the CPU with
numpy, ``ctc_prefix_score.py``; an on-device scorer is the remaining part of §8f row 1).
"""
import math
from typing import Dict, List, Optional

import torch


class SequenceGenerator:
    def __init__(self, models, tgt_dict, beam_size=1, max_len_a=0, max_len_b=200, min_len=1, normalize_scores=True,
                 len_penalty=1.0, unk_penalty=0.0, temperature=1.0, match_source_len=False, no_repeat_ngram_size=0,
                 search_strategy=None, eos=None, symbols_to_strip_from_output=None, lm_model=None, lm_weight=1.0,
                 ctc_weight=0.0):
        models = list(models) if isinstance(models, (list, tuple)) else [models]
        if len(models) != 1:
            raise NotImplementedError("model ensembles")
        if match_source_len or no_repeat_ngram_size or search_strategy is not None or lm_model is not None or ctc_weight:
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

        for sent in range(bsz):
            finalized[sent].sort(key=lambda h: -float(h["score"]))
        return finalized
