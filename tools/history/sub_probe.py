import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from s2t_amd import kernels as K
dev = "cuda"
torch.manual_seed(0)
def run(Bz, Tp, Cin, Cout, Kw=5, stride=2):
    Tout = (Tp + 2 * 2 - Kw) // stride + 1
    x = torch.randn(Bz, (Tp + 8) * Cin, device=dev).to(torch.bfloat16)
    W = (torch.randn(Cout, Kw * Cin, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev)
    out = torch.empty(Bz, Tout, Cout // 2, dtype=torch.bfloat16, device=dev)
    res = []
    for mode in (0, 2, 1):
        K.gemm_configure(mode)
        ts = []
        for r in range(11):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                K.gemm(x, W, out, M=Tout, N=Cout, K=Kw * Cin, lda=stride * Cin, ldb=Kw * Cin, ldc=Cout // 2, batch=Bz,
                       a_s=((Tp + 8) * Cin, 0), c_s=(Tout * (Cout // 2), 0), bias=b, act="glu")
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 250)
        res.append(sorted(ts)[5])
    print("B%d T%d Cin%d Cout%d : old %.1f  forced %.1f  auto %.1f us" % (Bz, Tp, Cin, Cout, *res))
run(64, 1000, 80, 1024); run(64, 500, 512, 512); run(256, 1000, 80, 4096); run(256, 500, 2048, 1024); run(64, 2000, 80, 1024)
