import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_model_parity_gpu import build, load
from s2t_amd import criterions as C, s2t_transformer as M
name = sys.argv[1]
z = load(os.path.join("tests", "golden"), name)
for dtype in (torch.float32, torch.bfloat16):
    model, cfg = build(z, dtype)
    model.train()
    crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(M.FakeTask(model.decoder.output_projection.weight.shape[0]), label_smoothing=0.1, ctc_weight=cfg["ctc_weight"])
    sample = {"net_input": {"src_tokens": torch.from_numpy(z["in::src_tokens"]).cuda(), "src_lengths": torch.from_numpy(z["in::src_lengths"]).cuda(),
              "prev_output_tokens": torch.from_numpy(z["in::prev_output_tokens"]).cuda()}, "target": torch.from_numpy(z["in::target"]).cuda(), "ntokens": int(z["in::ntokens"])}
    model.flat.zero_grad()
    loss, _, log = crit(model, sample); loss.backward(); torch.cuda.synchronize()
    errs = []
    for k in z.files:
        if k.startswith("grad::"):
            key = k[6:]; ref = z[k]; g = dict(model.named_parameters())[key].grad.float().cpu().numpy()
            if "subsample" in key and ref.ndim == 3: g = g.transpose(0, 2, 1)
            errs.append((np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-3), np.linalg.norm(g - ref) / max(np.linalg.norm(ref), 1e-6), key))
    errs.sort(reverse=True)
    print(dtype, "loss", log["loss"], float(z["out::loss"]))
    for e in errs[:12]: print("  max-rel %.4f  l2-rel %.4f  %s" % e)
