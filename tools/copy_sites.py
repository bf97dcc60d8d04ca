#!/usr/bin/env python3
"""Where the `__amd_rocclr_copyBuffer` launches come from: every copy that crosses the host / device
boundary or moves device bytes through hipMemcpyAsync, of ANY size, by the s2t_amd frames that issued it — for

  enc    one eager eval pass of the headline encoder (what tools/enc_fwd_profile.py traces),
  step   one eager training step,
  replay one captured replay with a NEW batch handed in (Trainer.replay(sample): load_batch + bookkeeping refresh + graph launch):
         what runs OUTSIDE the graph per step.  Copies recorded during the capture itself are graph nodes and are listed apart.

TorchDispatchMode sees ATen ops only; the library's own launches (ctypes) and hipMemcpy calls it makes are not ATen ops — the
C side has no hipMemcpy at all (grep csrc/), and the ctypes side's only transfers are the ones kernels.py issues through torch.
usage: python tools/copy_sites.py [enc] [step] [replay]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402
from s2t_amd import criterions as C, s2t_transformer as M  # noqa: E402
from s2t_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
V = 10000
task = M.FakeTask(V)
COPY_OPS = ("aten.copy_", "aten._to_copy", "aten.clone", "aten.contiguous", "aten.cat", "aten._local_scalar_dense", "aten.item",
            "aten.lift_fresh", "aten.scalar_tensor", "aten.fill_", "aten.zero_", "aten.zeros", "aten.full", "aten.arange", "aten.index_put")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(COPY_OPS):
            ts = [a for a in list(args) + [out] if torch.is_tensor(a)]
            devs = sorted({t.device.type for t in ts})
            kind = "+".join(devs) if devs else "scalar"
            if name.startswith(("aten._local_scalar_dense", "aten.item")):
                kind = "D2H(sync)"
            elif len(devs) == 2:
                kind = "H<->D"
            n = max([t.numel() * t.element_size() for t in ts], default=0)
            st = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack() if "s2t_amd" in f.filename][-3:]
            self.seen[(kind, name.split(".default")[0], n, " <- ".join(reversed(st)))] += 1
        return out

    def report(self, title):
        print("== %s: %d copy-like ATen ops" % (title, sum(self.seen.values())))
        by_kind = collections.Counter()
        for (kind, name, n, st), c in self.seen.items():
            by_kind[kind] += c
        print("   by kind:", dict(by_kind))
        for (kind, name, n, st), c in sorted(self.seen.items(), key=lambda kv: (kv[0][0], -kv[1])):
            print(f"   x{c:3d} {kind:10s} {name:26s} {n:>11d} B  {st}")


which = sys.argv[1:] or ["enc", "step", "replay"]
torch.manual_seed(1)
m = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V, dropout=0.1, attention_dropout=0.1, activation_dropout=0.1),
                                      task).prepare(torch.bfloat16, dev)
sample, frames = bench.synthetic_batch(64, 1000, V, 1, dev)
sample2, _ = bench.synthetic_batch(64, 1000, V, 2, dev)
ni = sample["net_input"]
if "enc" in which:
    m.eval()
    with torch.no_grad():
        for _ in range(2):
            m.encoder(ni["src_tokens"], ni["src_lengths"])
        torch.cuda.synchronize()
        with Spy() as spy:
            m.encoder(ni["src_tokens"], ni["src_lengths"])
        torch.cuda.synchronize()
    spy.report("one eager eval encoder pass (same batch object as the pass before)")
    with torch.no_grad(), Spy() as spy:
        m.encoder(sample2["net_input"]["src_tokens"], sample2["net_input"]["src_lengths"])
    torch.cuda.synchronize()
    spy.report("one eager eval encoder pass over a NEW batch object (per-batch bookkeeping recomputed)")
if "step" in which or "replay" in which:
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(task, label_smoothing=0.1, ctc_weight=0.3)
    tr = Trainer(m, crit)
    for _ in range(2):
        tr.train_step(sample)
    torch.cuda.synchronize()
    if "step" in which:
        with Spy() as spy:
            tr.train_step(sample)
        torch.cuda.synchronize()
        spy.report("one eager training step")
    if "replay" in which:
        with Spy() as spy:
            tr.capture(sample)
        torch.cuda.synchronize()
        spy.report("Trainer.capture (2 eager warm-up steps + the captured pass: its copies become graph nodes)")
        tr.replay(sample2)
        torch.cuda.synchronize()
        with Spy() as spy:
            tr.replay(sample)
        torch.cuda.synchronize()
        spy.report("one replay with a new batch handed in: everything listed here runs OUTSIDE the graph")
