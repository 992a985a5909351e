#!/usr/bin/env python3
"""The six stride-2 layers of ResNet-50 at the bench batch (64 frames of 256x256): conv2 (3x3 / s2) and the downsample unit
(1x1 / s2) of the first block of layers 2-4 - forward, dgrad, weight gradient (kernel + reduction) through the C ABI.
tools/bench_s2.py [opt=value ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd._lib import get_lib  # noqa: E402
from vfs_amd.packing import wgrad_splits  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout, k, stride, pad
    (64, 64, 64, 128, 128, 3, 2, 1), (64, 32, 32, 256, 256, 3, 2, 1), (64, 16, 16, 512, 512, 3, 2, 1),
    (64, 64, 64, 256, 512, 1, 2, 0), (64, 32, 32, 512, 1024, 1, 2, 0), (64, 16, 16, 1024, 2048, 1, 2, 0),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    lib = get_lib()
    for kv in [a for a in sys.argv[1:] if '=' in a]:
        k, v = kv.split('=')
        lib.set_option(k.encode(), int(v))
    dev = torch.device('cuda:0')
    s = torch.cuda.current_stream().cuda_stream
    print(' '.join(sys.argv[1:]) or 'defaults')
    tot = [0.0, 0.0, 0.0]
    for (N, H, W, Cin, Cout, k, st, pad) in SHAPES:
        Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        M = N * Ho * Wo
        x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
        wf = torch.randn(Cout, k, k, Cin, device=dev).to(torch.bfloat16)
        wd = torch.randn(Cin, k, k, Cout, device=dev).to(torch.bfloat16)
        y = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
        dy = torch.randn(N, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
        dx = torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
        stats = torch.empty((M + 127) // 128 * 2 * Cout, device=dev)
        nsplit, pps = wgrad_splits(M, Cout, k * k * Cin)
        partial = torch.empty(nsplit * Cout * k * k * Cin, device=dev)
        grad = torch.zeros(Cout, Cin, k, k, device=dev)
        fl = 2.0 * M * Cout * k * k * Cin
        tf = timeit(lambda: lib.conv_fwd(x, wf, y, None, stats, N, H, W, Cin, Ho, Wo, Cout, k, k, st, pad, s))
        td = timeit(lambda: lib.conv_dgrad(dy, wd, dx, None, N, H, W, Cin, Ho, Wo, Cout, k, k, st, pad, s))
        tw = timeit(lambda: lib.conv_wgrad(dy, x, partial, grad, N, H, W, Cin, Ho, Wo, Cout, k, k, st, pad, nsplit, pps, s))
        tot[0] += tf; tot[1] += td; tot[2] += tw
        print(f'{str((N, H, W, Cin, Cout, k, st)):36s} fwd {tf * 1e6:6.1f} us {fl / tf / 1e12:5.0f} TF/s | dgrad {td * 1e6:6.1f} us {fl / td / 1e12:5.0f} TF/s | '
              f'wgrad {tw * 1e6:6.1f} us {fl / tw / 1e12:5.0f} TF/s (nsplit {nsplit})', flush=True)
    print('sum us: fwd %.1f dgrad %.1f wgrad %.1f' % tuple(t * 1e6 for t in tot))


if __name__ == '__main__':
    main()
