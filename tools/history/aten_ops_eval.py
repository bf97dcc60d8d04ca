#!/usr/bin/env python3
"""Which ATen (non-libs2t_hip) GPU kernels and copies one eval-mode encoder forward still launches, by call site."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from s2t_amd import s2t_transformer as M
dev = torch.device("cuda", 0); V = 10000; task = M.FakeTask(V)
m = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V), task).prepare(torch.bfloat16, dev)
m.eval()
sample, frames = bench.synthetic_batch(64, 1000, V, 1, dev)
ni = sample["net_input"]
with torch.no_grad():
    for _ in range(3): m.encoder(ni["src_tokens"], ni["src_lengths"])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        m.encoder(ni["src_tokens"], ni["src_lengths"])
        torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        st = [s for s in e.stack if "s2t_amd" in s or "bench.py" in s]
        rows.append((e.device_time_total, e.count, e.key, " <- ".join(x[-48:] for x in st[:2])))
for t, c, k, s in sorted(rows, reverse=True)[:40]:
    print("%8.1f us  x%3d  %-22s %s" % (t, c, k, s))
