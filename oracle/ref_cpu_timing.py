#!/usr/bin/env python3
"""Time the REFERENCE's own PyTorch-CPU path (build container only; TEST INFRASTRUCTURE — the reference never travels).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo/oracle python /root/repo/oracle/ref_cpu_timing.py

The headline model (s2t_transformer_s + conformer.yaml + ctc.yaml, 12 enc / 6 dec, V = 10000) built by the reference's
``build_model``, the reference's joint criterion, forward + backward on synthetic batches of B x 1000 x 80 frames;
BASELINE.md §3 protocol: 3 warm-ups, 10 timed iterations, median.  Prints frames/s and the thread count."""
import statistics
import sys
import time

import ref_stubs

ref_stubs.install()
import torch  # noqa: E402

import gen_golden as G  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
WARM, ITERS = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 10)
V, T = 10000, 1000
conf = dict(macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, encoder_attention_type="rel_pos",
            encoder_activation_fn="swish", layer_padding_mask=True)
torch.manual_seed(1)
model, args, task = G.build("s2t_transformer_s", V, encoder_layers=12, decoder_layers=6, dropout=0.1, attention_dropout=0.1,
                            activation_dropout=0.1, **conf)
crit = G.criterion_for(task, args)
model.train()
crit.train()
src, lens, prev, target, ntokens = G.make_batch(B, T, V, 7, umin=20, umax=60)
sample = {"id": torch.arange(B), "net_input": {"src_tokens": src, "src_lengths": lens, "prev_output_tokens": prev},
          "target": target, "ntokens": ntokens}
frames = int(lens.sum())


def step():
    model.zero_grad()
    loss, _, _ = crit(model, sample)
    loss.backward()


for _ in range(WARM):
    step()
ts = []
for _ in range(ITERS):
    t0 = time.perf_counter()
    step()
    ts.append(time.perf_counter() - t0)
med = statistics.median(ts)
print("reference CPU fwd+bwd: B=%d T=%d threads=%d  median %.2f s/iter over %d (+%d warm-ups)  %.1f frames/s" % (
    B, T, torch.get_num_threads(), med, ITERS, WARM, frames / med))
