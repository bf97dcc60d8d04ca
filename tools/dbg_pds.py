import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from s2t_amd import criterions as C, s2t_transformer as M, pdss2t_transformer as PDS
dev = torch.device("cuda", 0); V = 10000; task = M.FakeTask(V)
torch.manual_seed(1)
conformer = len(sys.argv) < 2 or sys.argv[1] != "plain"
drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
a = M.recipe_args(conformer=conformer, vocab_size=V, arch="pdss2t_transformer_s_8", pds_stages=4, pds_layers="3_3_3_3",
                  pds_ratios="2_2_1_2", pds_fusion=False, pds_embed_dims="256_256_256_256", pds_ds_method="conv",
                  pds_embed_norm=True, pds_position_embed="1_1_1_1", pds_kernel_sizes="5_5_5_5",
                  pds_ffn_ratios="8_8_8_8", pds_attn_heads="4_4_4_4", dropout=drop, attention_dropout=drop, activation_dropout=drop)
m = PDS.PDSS2TTransformerModel.build_model(a, task).prepare(torch.bfloat16, dev)
m.train()
sample, frames = bench.synthetic_batch(64, T, V, 1, dev)
crit = C.LabelSmoothedCrossEntropyCriterionWithCTC(task, label_smoothing=0.1, ctc_weight=0.3)
enc = m.encoder
hooks = []
def mk(name):
    def h(mod, inp, out):
        t = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(t):
            print("%-40s finite=%s absmax=%.3g" % (name, bool(torch.isfinite(t.float()).all()), float(t.float().abs().max())))
    return h
for n, mod in enc.named_modules():
    if n.count(".") <= 1 and n:
        hooks.append(mod.register_forward_hook(mk(n)))
m.flat.zero_grad()
loss, ss, log = crit(m, sample)
print("loss", float(loss), {k: float(v) if not isinstance(v, int) else v for k, v in log.items() if "loss" in k})
for h in hooks: h.remove()
from s2t_amd.trainer import Trainer
tr = Trainer(m, crit)
for step in range(40):
    loss, log = tr.train_step(sample)
    torch.cuda.synchronize()
    g = m.flat.grad
    bad = [n for n, p in m.named_parameters() if not bool(torch.isfinite(p.grad).all())]
    print("step", step, "loss %.1f" % float(loss), "gnorm", float(tr.hyper[3]), "nonfinite grads:", len(bad), bad[:6],
          "master finite:", bool(torch.isfinite(m.flat.master).all()))
    if bad:
        break
