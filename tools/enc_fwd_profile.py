"""Encoder forward only (eval, headline configuration), N eager passes — for `rocprofv3 --kernel-trace --stats`.
usage: enc_fwd_profile.py [passes=10] [batch=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2t_amd import s2t_transformer as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
torch.manual_seed(1)
V = 10000
model = M.S2TTransformerModel.build_model(M.recipe_args(conformer=True, vocab_size=V), M.FakeTask(V)).prepare(torch.bfloat16, dev)
model.eval()
sample, frames = bench.synthetic_batch(B, 1000, V, 1, dev)
ni = sample["net_input"]
with torch.no_grad():
    for _ in range(n):
        model.encoder(ni["src_tokens"], ni["src_lengths"])
torch.cuda.synchronize()
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(n):
        model.encoder(ni["src_tokens"], ni["src_lengths"])
torch.cuda.synchronize()
print("done: %d passes, batch %d x 1000 (%d frames), eager %.3f ms per pass" % (n, B, frames, (time.perf_counter() - t0) / n * 1e3))
