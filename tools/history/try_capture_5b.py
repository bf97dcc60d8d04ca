#!/usr/bin/env python3
"""Does the NAST recipe's training step (CtcCriterion with the ground-truth curriculum) capture into a hipGraph?"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2t_amd import criterions as C, s2t_transformer as M
from s2t_amd.trainer import Trainer
dev = torch.device("cuda", 0)
V = 10000
task = M.FakeTask(V)
nast = dict(encoder_type="sate", text_encoder_layers=12, acoustic_encoder="transformer", adapter="inter_league",
            xctc_weight=1.0, ctc_weight=1.0, share_ctc_and_embed=True, share_xctc_and_embed=True, text_no_pos_emb=True,
            textual_encoder_embed_norm=False, textual_encoder_no_scale_embedding=True, encoder_normalize_before=True,
            share_inter_ctc=True, inter_ctc_weight=1.0, inter_ctc_layers="6,9", inter_xctc_weight=1.0,
            inter_xctc_layers="6,9", ctc_pae="inter_league", xctc_pae="inter_league", xctc_cross_attn=True,
            cross_attn_start_layer=4, cross_attn_layer=3, cross_attn_collaboration_mode="serial",
            cross_attn_league_drop_net=True, cross_attn_league_drop_net_prob=0.1, xctc_pae_ground_truth_ratio=0.8,
            xctc_pae_ground_truth_only_mistake=True, pae_oracle_smooth=True, encoder_embed_dim=512,
            encoder_ffn_embed_dim=2048, encoder_attention_heads=8, subsampling_filter=2048, activation_fn="relu")
a = M.recipe_args(conformer=True, vocab_size=V, dropout=0.15, attention_dropout=0.15, activation_dropout=0.15, **nast)
torch.manual_seed(1)
m = M.S2TCTCModel.build_model(a, task).prepare(torch.bfloat16, dev)
m.train()
crit = C.CtcCriterion(None, task, ctc_weight=1.0, inter_ctc_weight=1.0, xctc_weight=1.0, inter_xctc_weight=1.0)
crit.train()
tr = Trainer(m, crit)
sample, frames = bench.synthetic_batch(64, 1000, V, 1, dev)
sample["transcript"] = {"tokens": sample["target"]}
out = tr.train_step(sample)
torch.cuda.synchronize()
print("eager ok, loss", float(out[0]), flush=True)
try:
    tr.capture(sample)
    for _ in range(2):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = tr.replay()
    torch.cuda.synchronize()
    print("captured: %.2f ms/step loss %.1f" % ((time.perf_counter() - t0) / 5 * 1e3, float(out[0])), flush=True)
except Exception:
    traceback.print_exc()
