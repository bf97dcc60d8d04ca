"""Encoder forward only (eval, headline configuration), N eager passes — for `rocprofv3 --kernel-trace --stats`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2t_amd import s2t_transformer as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(1)
V = 10000
model = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V), M.FakeTask(V)).prepare(torch.bfloat16, dev)
model.eval()
sample, frames = bench.synthetic_batch(64, 1000, V, 1, dev)
ni = sample["net_input"]
with torch.no_grad():
    for _ in range(n):
        model.encoder(ni["src_tokens"], ni["src_lengths"])
torch.cuda.synchronize()
print("done", n)
