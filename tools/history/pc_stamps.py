"""In-kernel s_memtime stamps of the producer / consumer FFN kernel (library built with -DS2T_PC_DBG=16, tools/pc_dbg_build.sh)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import _lib as L
from s2t_amd import kernels as K

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
ws = K._ffn_pair_ws(a, M, x.device)
if train:
    a.z, a.h = z.data_ptr(), h.data_ptr()
    a.z_tiled_ok = 1
    a.drop_h_p, a.drop_h_site, a.drop_o_p, a.drop_o_site, a.drop_seed = 0.1, 1, 0.1, 2, seed.data_ptr()
for _ in range(3):
    L.check(L.lib().s2t_ffn_fused_fwd(C.byref(a), L.stream_ptr()), "ffn")
torch.cuda.synchronize()
full = dbg.cpu()[:256].view(2, 8, 16)
names = ["prologue", "frag+barrier", "loop", "sync", "y->lds", "send", "wait", "epilogue"]
for blk in range(2):
    for w in range(8):
        s = [int(v) for v in full[blk, w, :8]]
        dl = [s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[6] - s[5], s[7] - s[6]]
        real = int(full[blk, w, 8])
        print("block %3d wave %d (%s): pro %6d loop %6d sync %5d y->lds %5d send %5d wait %5d epi %5d | total %6d cycles, %.1f us -> %.0f MHz"
              % ([0, 100][blk], w, "P" if w < 4 else "C", dl[0], dl[1], dl[2], dl[3], dl[4], dl[5], dl[6], s[7] - s[0], real / 100.0,
                 (s[7] - s[0]) / (real / 100.0)))
        sg = [int(v) for v in full[blk, w, 9:13]]
        print("      loop segments (sum over chunks): " + " ".join("%6d" % v for v in sg) + ("   [tileA | tileB+E1a | E1b | barrier]" if w < 4 else "   [nt0-3 | nt4-7 | wait | barrier]"))
