import torch
torch.cuda.init()
x = torch.zeros(1, device="cuda")
torch.cuda.synchronize()
pairs = []
for _ in range(200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); pairs.append((e0, e1))
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
print("empty event pair us: min %.2f median %.2f p90 %.2f" % (t[0], t[100], t[180]))
