#!/usr/bin/env python3
"""s2t_conv_bwd_fused at the bench's shape (64 utterances x 250 frames, 256 channels, kernel 15), us per call (HIP events, 50 calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2t_amd import kernels as K
dev = "cuda"; B, T, C, Kw = 64, 250, 256, 15
n = B * T
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
D, dA, G = (r(n, C).bfloat16() for _ in range(3)); Z = r(n, 2 * C).bfloat16(); dZ = torch.empty_like(Z)
w = r(C, Kw) * 0.2; scale, shift, mean = r(C), r(C), r(C); rstd = r(C).abs() + 0.5; sums = r(2 * C); dw = torch.zeros(C, Kw, device=dev)
lens = torch.randint(150, T + 1, (B,), device=dev, dtype=torch.int32)
def run(): K.conv_bwd_fused(D, dA, G, Z, w, scale, shift, mean, rstd, sums, float(n), "swish", lens, dZ, dw, B, T, C, Kw, defer_slot=0)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("conv_bwd_fused %d x %d x %d, K = %d: %.1f us per call" % (B, T, C, Kw, e0.elapsed_time(e1) / 50 * 1e3))
# the forward depthwise convolution with the BatchNorm statistics (training form)
x = r(n, C).bfloat16(); y = torch.empty_like(x); stats = torch.empty(K.dwconv_stat_partials(B, T), 2, C, device=dev)
def fwd(): K.dwconv_fwd(x, w, y, B, T, C, Kw, stats=stats)
for _ in range(5): fwd()
torch.cuda.synchronize()
e0.record()
for _ in range(50): fwd()
e1.record(); torch.cuda.synchronize()
print("dwconv_fwd + statistics: %.1f us per call" % (e0.elapsed_time(e1) / 50 * 1e3))
