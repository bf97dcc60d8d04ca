#!/usr/bin/env python3
"""Sanity: the headline configuration over-fits one fixed synthetic batch (loss per token falls) under hipGraph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2t_amd import criterions as C, s2t_transformer as M
from s2t_amd.trainer import Trainer
dev = torch.device("cuda", 0)
V = 10000
torch.manual_seed(1)
m = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V, dropout=0.1, attention_dropout=0.1,
                                                    activation_dropout=0.1), M.FakeTask(V)).prepare(torch.bfloat16, dev)
crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
tr = Trainer(m, crit)
tr.warmup_updates = 200  # reach the peak learning rate quickly
sample, frames = bench.synthetic_batch(64, 1000, V, 1, dev)
tr.train_step(sample)
tr.capture(sample)
ntok = sample["ntokens"]
for i in range(1, 601):
    out = tr.replay()
    if i % 100 == 0 or i == 1:
        torch.cuda.synchronize()
        print("update %4d  loss/token %.4f  lr %.2e  gnorm %.3f" % (tr.num_updates, float(out[0]) / ntok, tr.lr_at(tr.num_updates), float(tr.hyper[3])), flush=True)
