#!/usr/bin/env python3
"""Achievable HBM write / copy bandwidth (torch fill / copy kernels) for sizing the GEMM epilogue cost."""
import torch
dev = "cuda"
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (16, 65, 130, 520):
    a = torch.empty(mb * 1000 * 1000 // 2, dtype=torch.bfloat16, device=dev)
    b = torch.empty_like(a)
    us = t(lambda: a.zero_())
    print("fill  %4d MB: %7.1f us  %6.2f TB/s (write)" % (mb, us, mb / us))
    us = t(lambda: b.copy_(a))
    print("copy  %4d MB: %7.1f us  %6.2f TB/s (read+write)" % (mb, us, 2 * mb / us))
    us = t(lambda: a.sum())
    print("sum   %4d MB: %7.1f us  %6.2f TB/s (read)" % (mb, us, mb / us))
