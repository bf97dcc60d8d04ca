#!/usr/bin/env python3
"""Per-(symbol, shape) GEMM timing of one instrumented training step (HIP events around every s2t_gemm launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2t_amd import criterions as C, kernels as K, s2t_transformer as M
from s2t_amd.trainer import Trainer

arch = sys.argv[1] if len(sys.argv) > 1 else "conformer"
dev = torch.device("cuda", 0)
V = 10000
torch.manual_seed(1)
model = M.S2TTransformerModel.build_model(M.recipe_args(conformer=arch == "conformer", vocab_size=V), M.FakeTask(V)).prepare(torch.bfloat16, dev)
crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
tr = Trainer(model, crit)
sample, frames = bench.synthetic_batch(64, 1000, V, 1, dev)
for _ in range(2):
    tr.train_step(sample)
torch.cuda.synchronize()
K.GEMM_PROFILE = []
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); tr.train_step(sample); t1.record()
torch.cuda.synchronize()
prof, K.GEMM_PROFILE = K.GEMM_PROFILE, None
agg = {}
for sym, fl, e0, e1, shape, *_ in prof:
    a = agg.setdefault((sym, shape), [0.0, 0.0, 0])
    a[0] += fl; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
tot = sum(a[1] for a in agg.values())
print("step %.2f ms, gemm %.2f ms" % (t0.elapsed_time(t1), tot * 1e3))
for (sym, shape), (fl, sec, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-38s M,N,K,b=%-24s n=%3d avg %7.1f us  tot %6.2f ms  %6.1f TF/s" % (sym[12:-1].replace("unsigned short", "bf16").replace("float", "f32"), shape, n, sec / n * 1e6, sec * 1e3, fl / sec / 1e12))
