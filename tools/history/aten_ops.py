#!/usr/bin/env python3
"""Which ATen (non-libs2t_hip) GPU kernels a training step still launches, with call stacks' top frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from s2t_amd import criterions as C, s2t_transformer as M
from s2t_amd.trainer import Trainer
dev = torch.device("cuda", 0); V = 10000; task = M.FakeTask(V)
m = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V, dropout=0.1, attention_dropout=0.1, activation_dropout=0.1), task).prepare(torch.bfloat16, dev)
crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(task, label_smoothing=0.1, ctc_weight=0.3)
tr = Trainer(m, crit)
sample, frames = bench.synthetic_batch(64, 1000, V, 1, dev)
for _ in range(3): tr.train_step(sample)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(sample)
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if e.name.startswith("aten::") and e.device_time_total > 0 and e.name not in ("aten::zero_", "aten::zeros", "aten::clone", "aten::contiguous", "aten::zeros_like", "aten::to", "aten::ones_like"):
        st = [s for s in (e.stack or []) if "s2t_amd" in s or "bench.py" in s]
        key = (e.name, " <- ".join(x.split("/")[-1][:60] for x in st[:3]), str(e.input_shapes)[:60])
        t, c = rows.get(key, (0.0, 0))
        rows[key] = (t + e.device_time_total, c + 1)
for (k, s, shp), (t, c) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:50]:
    print("%8.1f us  x%3d  %-22s %s %s" % (t, c, k, shp, s))
