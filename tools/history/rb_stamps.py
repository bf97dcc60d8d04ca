"""In-kernel s_memtime stamps of the last loop iteration of the fused FFN kernel (library built with -DS2T_RB_DBG=16)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import _lib as L

DEV = "cuda"
M, d, F = 16000, 256, 2048
g = torch.Generator().manual_seed(0)
x = torch.randn(M, d, generator=g).bfloat16().to(DEV)
w1 = (torch.randn(F, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
w2 = (torch.randn(d, F, generator=g) * F ** -0.5).bfloat16().to(DEV)
b1 = torch.zeros(F, device=DEV); b2 = torch.zeros(d, device=DEV)
gam = torch.ones(d, device=DEV); bet = torch.zeros(d, device=DEV)
y = torch.empty_like(x)
dbg = torch.zeros(512, dtype=torch.int64, device=DEV)
train = len(sys.argv) > 1 and sys.argv[1] == "train"
z = torch.empty(M, F, dtype=torch.bfloat16, device=DEV) if train else None
h = torch.empty(M, F, dtype=torch.bfloat16, device=DEV) if train else None
seed = torch.tensor([1], dtype=torch.int64, device=DEV)
a = L.FfnArgs()
a.x, a.d, a.M, a.F = x.data_ptr(), d, M, F
a.ln_gamma, a.ln_beta, a.ln_eps = gam.data_ptr(), bet.data_ptr(), 1e-5
a.w1, a.b1, a.w2, a.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
a.residual, a.y = x.data_ptr(), y.data_ptr()
a.eln_mean = dbg.data_ptr()
a.act, a.alpha = 2, 0.5
if train:
    a.z, a.h = z.data_ptr(), h.data_ptr()
    a.drop_h_p, a.drop_h_site, a.drop_o_p, a.drop_o_site, a.drop_seed = 0.1, 1, 0.1, 2, seed.data_ptr()
for _ in range(3):
    L.check(L.lib().s2t_ffn_fused_fwd(C.byref(a), L.stream_ptr()), "ffn")
torch.cuda.synchronize()
full = dbg.cpu()[:256].view(2, 8, 16)
t = full[:, :, :9]
for blk in range(2):
    for w in (0, 4):
        pro, loop, epi, real = (int(full[blk, w, i]) for i in (9, 10, 11, 12))
        tot = pro + loop + epi
        print("block %3d wave %d: prologue %6d  loop %6d  epilogue %6d cycles; %d ticks of 100 MHz -> %.0f MHz shader clock, kernel %.1f us"
              % ([0, 100][blk], w, pro, loop, epi, real, tot / (real / 100.0), real / 100.0))
per_it = dbg.cpu()[256:384].view(2, 64)[:, :32]
for blk in range(2):
    print("block %3d wave 0 cycles per iteration:" % [0, 100][blk], " ".join(str(int(v)) for v in per_it[blk]))
names = []
for blk in range(2):
    print("block", [0, 100][blk])
    for w in range(8):
        s = t[blk, w]
        dl = [int(s[i + 1] - s[i]) for i in range(8)]
        print("  wave %d: total %5d | " % (w, int(s[8] - s[0])) + " ".join("%5d" % v for v in dl))
print("nh=0: reads+dma_w1 | g1a | dma_w2 | g1b | e1 | g2a | g2b | wait+barrier;  nh=1: reads+dma_w1 | g2a | dma_w2 | g2b | g1a | g1b | e1 | wait+barrier")
