#!/usr/bin/env python3
"""Where do the small device-to-device copies of a config-5b greedy pass come from? (torch.profiler, grouped by Python stack)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2t_amd import s2t_transformer as M
from torch.profiler import profile, ProfilerActivity
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
m.encoder.xctc_out_dtype = torch.float32
m.eval()
sample, frames = bench.synthetic_batch(64, 1000, V, 2, dev)
dec = M.CTCDecoder([m], None, None)
with torch.no_grad():
    for _ in range(2):
        dec.generate([m], sample)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        dec.generate([m], sample)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60, max_src_column_width=110))
