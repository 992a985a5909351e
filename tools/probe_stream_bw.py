#!/usr/bin/env python3
"""What streaming bandwidth does this box deliver to the simplest possible kernels?  (the ceiling of the BatchNorm passes)"""
import torch
dev = torch.device('cuda:0')
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (33, 134, 537, 2147):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    tc = t(lambda: y.copy_(x))
    ta = t(lambda: torch.add(x, y, out=z))
    tr = t(lambda: x.float().sum()) if mb <= 537 else float('nan')
    print(f'{mb:5d} MB tensors: copy (1R+1W) {2 * mb / tc / 1e6:6.2f} TB/s | add (2R+1W) {3 * mb / ta / 1e6:6.2f} TB/s')
