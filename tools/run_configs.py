#!/usr/bin/env python3
"""The BASELINE.json configurations at their full sizes on one MI355X (sanity + timings; bench.py reports config 2').

  1   s2t_transformer_s 6/6, 32 x 400 x 80, fp32, eval + CTC greedy
  2   s2t_transformer_s 12/6 Transformer, 64 x 1000 x 80, bf16, training step
  2'  + Conformer (= bench.py default)
  3   pdss2t_transformer_s_8 (+ Conformer flags), 64 x 2000 x 80, bf16, training step
  4   s2t_sate 12+6 / 6, 64 x 1000 x 80, bf16, training step (per-GPU share of the 8-GPU config)
  5a  s2t_ctc 12L Conformer encoder, 256 x 1000 x 80, bf16 eval, CTC greedy
  5b  the full reproduction_nast.yaml stack (s2t_ctc --encoder-type sate: 12L Conformer acoustic + 12 textual layers with
      cross-layer attention, d=512/h=8/F=2048, inter-(X)CTC at 6,9 with PAE), 256 x 1000 x 80 bf16 eval, greedy on
      xctc_logit; plus one eager training step of the same stack (CtcCriterion with the ground-truth curriculum) at 64 x 1000
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2t_amd import criterions as C, s2t_transformer as M, pdss2t_transformer as PDS, s2t_sate as SATE
from s2t_amd.trainer import Trainer

dev = torch.device("cuda", 0)
V = 10000
task = M.FakeTask(V)
drop = dict(dropout=0.1, attention_dropout=0.1, activation_dropout=0.1)
conf = dict(macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, encoder_attention_type="rel_pos",
            encoder_activation_fn="swish", layer_padding_mask=True)


def train_cfg(name, model, B, T, steps=8):
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(task, label_smoothing=0.1, ctc_weight=0.3)
    tr = Trainer(model, crit)
    sample, frames = bench.synthetic_batch(B, T, V, 1, dev)
    tr.train_step(sample)
    tr.capture(sample)
    for _ in range(2):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("%-4s train %4d x %4d : %7.2f ms/step  %7.3f M frames/s  loss %.1f  peak mem %.1f GB" % (
        name, B, T, dt * 1e3, frames / dt / 1e6, float(out[0]), torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
    del tr
    torch.cuda.empty_cache()


def greedy_cfg(name, model, B, T, dtype):
    model.eval()
    sample, frames = bench.synthetic_batch(B, T, V, 2, dev)
    ni = sample["net_input"]
    dec = M.CTCDecoder([model], None, None) if hasattr(M, "CTCDecoder") else None
    with torch.no_grad():
        for _ in range(2):
            hyps = dec.generate([model], sample)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            hyps = dec.generate([model], sample)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    n = sum(len(h[0]["tokens"]) for h in hyps)
    print("%-4s greedy %3d x %4d %s : %7.2f ms/batch  %7.3f M frames/s  (%d tokens)" % (
        name, B, T, str(dtype).split(".")[-1], dt * 1e3, frames / dt / 1e6, n), flush=True)


which = sys.argv[1:] or ["1", "2", "2p", "3", "4", "5a"]
torch.manual_seed(1)
if "1" in which:
    a = M.recipe_args(conformer=False, vocab_size=V, encoder_layers=6, ctc_weight=1.0)
    m = M.S2TCTCModel.build_model(a, task).prepare(torch.float32, dev)
    greedy_cfg("1", m, 32, 400, torch.float32)
if "2" in which:
    m = M.S2TTransformerModel.build_model(M.recipe_args(conformer=False, vocab_size=V, **drop), task).prepare(torch.bfloat16, dev)
    train_cfg("2", m, 64, 1000)
if "2p" in which:
    m = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V, **drop), task).prepare(torch.bfloat16, dev)
    train_cfg("2'", m, 64, 1000)
if "3" in which:
    a = M.recipe_args(conformer=True, vocab_size=V, arch="pdss2t_transformer_s_8", pds_stages=4, pds_layers="3_3_3_3",
                      pds_ratios="2_2_1_2", pds_fusion=False, pds_embed_dims="256_256_256_256", pds_ds_method="conv",
                      pds_embed_norm=True, pds_position_embed="1_1_1_1", pds_kernel_sizes="5_5_5_5",
                      pds_ffn_ratios="8_8_8_8", pds_attn_heads="4_4_4_4", **drop)
    m = PDS.PDSS2TTransformerModel.build_model(a, task).prepare(torch.bfloat16, dev)
    train_cfg("3", m, 64, 2000)
if "4" in which:
    a = M.recipe_args(conformer=False, vocab_size=V, arch="s2t_sate", text_encoder_layers=6, acoustic_encoder="transformer",
                      adapter="inter_league", textual_encoder_embed_norm=True, textual_encoder_no_scale_embedding=True,
                      encoder_normalize_before=True, decoder_normalize_before=True, **drop)
    m = SATE.S2TSATEModel.build_model(a, task).prepare(torch.bfloat16, dev)
    train_cfg("4", m, 64, 1000)
if "5a" in which:
    a = M.recipe_args(conformer=True, vocab_size=V, ctc_weight=1.0)
    m = M.S2TCTCModel.build_model(a, task).prepare(torch.bfloat16, dev)
    m.encoder.ctc_out_dtype = torch.float32
    greedy_cfg("5a", m, 256, 1000, torch.bfloat16)
if "5b" in which or "5bg" in which:  # 5bg: the greedy pass only (profiling)
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
    m = M.S2TCTCModel.build_model(a, task).prepare(torch.bfloat16, dev)
    print("5b   params %.1f M" % (m.flat.master.numel() / 1e6), flush=True)
    m.encoder.xctc_out_dtype = torch.float32  # greedy decodes xctc_logit: fp32 there (bit-exact arg-max), compute dtype elsewhere
    greedy_cfg("5b", m, 256, 1000, torch.bfloat16)
    m.encoder.xctc_out_dtype = None
    if "5b" not in which:
        sys.exit(0)
    m.encoder.acoustic_encoder.ctc_out_dtype = None
    m.train()
    crit = C.CtcCriterion(None, task, ctc_weight=1.0, inter_ctc_weight=1.0, xctc_weight=1.0, inter_xctc_weight=1.0)
    crit.train()
    tr = Trainer(m, crit)
    sample, frames = bench.synthetic_batch(64, 1000, V, 1, dev)
    sample["transcript"] = {"tokens": sample["target"]}
    for _ in range(2):
        out = tr.train_step(sample)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        out = tr.train_step(sample)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print("5b   train (eager, 2 encoder passes: alignment oracle + curriculum)  64 x 1000 : %7.2f ms/step  %7.3f M frames/s  loss %.1f  peak mem %.1f GB" % (
        dt * 1e3, frames / dt / 1e6, float(out[0]), torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
    # the same step as ONE hipGraph (the alignment oracle stays on the device: torch_imputer.best_alignment_states)
    tr.capture(sample)
    for _ in range(2):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        out = tr.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    print("5b   train (hipGraph replay)  64 x 1000 : %7.2f ms/step  %7.3f M frames/s  loss %.1f" % (dt * 1e3, frames / dt / 1e6, float(out[0])), flush=True)
