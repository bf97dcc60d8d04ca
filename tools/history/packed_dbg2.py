"""Debug aid: eager vs captured packed steps over batches of different fill."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_packed_rows_gpu as T
from s2t_amd import criterions as C, s2t_transformer as M, trainer as TR

V = T.V
batches = [T._sample(24, 1000, 11 + i, full_first=(i % 2 == 0), lo=0.5 + 0.1 * i)[0] for i in range(3)]
seq = [0, 1, 2, 2, 2, 0, 1, 2, 0, 0]
for mode in sys.argv[1:] or ["padded_eager", "packed_eager", "packed_graph", "padded_graph"]:
    model = T._model(True, dropout=0.0)
    model.train()
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(V), label_smoothing=0.1, ctc_weight=0.3)
    with T._layout(mode.startswith("packed")):
        tr = TR.Trainer(model, crit, lr=1e-5, warmup_updates=1, clip_norm=10.0)
        out = {}
        if mode.endswith("eager"):
            for i, bi in enumerate(seq):
                l, log = tr.train_step(batches[bi])
                out[i] = (float(l), float(log["ctc_loss"]))
        else:
            for i in (0, 1):
                l, log = tr.train_step(batches[seq[i]])
                out[i] = (float(l), float(log["ctc_loss"]))
            tr.capture(batches[2])
            l, log = tr.replay()
            out[4] = (float(l), float(log["ctc_loss"]))
            for i in range(5, len(seq)):
                l, log = tr.replay(batches[seq[i]])
                out[i] = (float(l), float(log["ctc_loss"]))
        torch.cuda.synchronize()
    print(mode, " ".join("%d:%.1f/%.1f" % (i, a, b) for i, (a, b) in sorted(out.items())))
