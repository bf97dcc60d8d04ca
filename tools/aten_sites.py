#!/usr/bin/env python3
"""Where the ATen kernels of a training step come from: every aten op on >= MIN elements (argv[1], default 1000; 1 lists every op that
launches a kernel, the 4-byte fills included) with the s2t_amd frames that issued it (TorchDispatchMode; the profiler on this image
returns no Python stacks)."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from s2t_amd import criterions as C, s2t_transformer as M
from s2t_amd.trainer import Trainer
dev = torch.device("cuda", 0); V = 10000; task = M.FakeTask(V)
m = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V, dropout=0.1, attention_dropout=0.1, activation_dropout=0.1), task).prepare(torch.bfloat16, dev)
crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(task, label_smoothing=0.1, ctc_weight=0.3)
tr = Trainer(m, crit)
sample, frames = bench.synthetic_batch(64, 1000, V, 1, dev)
for _ in range(3): tr.train_step(sample)
torch.cuda.synchronize()
seen = collections.Counter()
MIN = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.as_strided", "aten.t.", "aten.transpose", "aten.select.", "aten.slice.", "aten.unsqueeze",
        "aten.squeeze", "aten.expand", "aten.permute", "aten.reshape", "aten.empty", "aten.new_empty", "aten.unbind", "aten.split", "aten._local_scalar")
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            ts = [a for a in list(args) + [out] if torch.is_tensor(a) and a.is_cuda]
            n = max([t.numel() for t in ts], default=0)
            if n >= MIN:
                st = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack() if "s2t_amd" in f.filename][-3:]
                seen[(name, tuple(ts[0].shape) if ts else (), " <- ".join(reversed(st)))] += 1
        return out
with Spy():
    tr.train_step(sample)
torch.cuda.synchronize()
for (name, shp, st), c in sorted(seen.items(), key=lambda kv: kv[0][2]):
    print(f"x{c:2d} {name:28s} {str(shp):22s} {st}")
