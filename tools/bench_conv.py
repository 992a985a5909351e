#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernels (fwd / dgrad / wgrad) at the bench shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd._lib import VfsLib, get_lib, set_lib  # noqa: E402
from vfs_amd.packing import wgrad_halo_eligible, wgrad_splits  # noqa: E402

R18 = [  # N, H, W, Cin, Cout, k, stride, pad   (N = 256 frames: 2 views x 32 videos x 4 frames)
    (256, 64, 64, 64, 64, 3, 1, 1), (256, 64, 64, 64, 128, 3, 2, 1), (256, 32, 32, 128, 128, 3, 1, 1),
    (256, 64, 64, 64, 128, 1, 2, 0), (256, 32, 32, 128, 256, 3, 2, 1), (256, 16, 16, 256, 256, 3, 1, 1),
    (256, 16, 16, 256, 512, 3, 2, 1), (256, 8, 8, 512, 512, 3, 1, 1),
]
R50 = [  # N = 64 frames
    (64, 64, 64, 64, 64, 1, 1, 0), (64, 64, 64, 64, 256, 1, 1, 0), (64, 64, 64, 256, 64, 1, 1, 0),
    (64, 64, 64, 128, 128, 3, 2, 1), (64, 32, 32, 128, 512, 1, 1, 0), (64, 32, 32, 512, 128, 1, 1, 0),
    (64, 16, 16, 256, 256, 3, 1, 1), (64, 16, 16, 256, 1024, 1, 1, 0), (64, 16, 16, 1024, 256, 1, 1, 0),
    (64, 8, 8, 512, 512, 3, 1, 1), (64, 8, 8, 512, 2048, 1, 1, 0), (64, 8, 8, 2048, 512, 1, 1, 0),
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
    if os.environ.get('VFS_HIP_LIB'):      # A/B a variant build of the library
        set_lib(VfsLib(os.environ['VFS_HIP_LIB']))
    lib = get_lib()
    dev = torch.device('cuda:0')
    s = torch.cuda.current_stream().cuda_stream
    which = sys.argv[1] if len(sys.argv) > 1 else 'r18'
    print(f'{"shape":44s} {"fwd TF/s":>9s} {"dgrad":>9s} {"wgrad":>9s}   ms(fwd/dgrad/wgrad)')
    tot = [0.0, 0.0, 0.0]
    for (N, H, W, Cin, Cout, k, st, pad) in (R18 if which == 'r18' else R50):
        Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        M = N * Ho * Wo
        x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
        wf = torch.randn(Cout, k, k, Cin, device=dev).to(torch.bfloat16)
        wd = torch.randn(Cin, k, k, Cout, device=dev).to(torch.bfloat16)
        y = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
        dy = torch.randn(N, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
        dx = torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
        stats = torch.empty((M + 127) // 128 * 2 * Cout, device=dev)
        halo = (N, H, W, Cin) if wgrad_halo_eligible(N, H, W, Cin, Cout, k, st, pad) else None
        nsplit, pps = wgrad_splits(M, Cout, k * k * Cin, halo_geom=halo)
        nsplit0, pps0 = wgrad_splits(M, Cout, k * k * Cin)
        partial0 = torch.empty(nsplit0 * Cout * k * k * Cin, device=dev)
        partial = torch.empty(nsplit * Cout * k * k * Cin, device=dev)
        grad = torch.zeros(Cout, Cin, k, k, device=dev)
        fl = 2.0 * M * Cout * k * k * Cin
        lib.set_option(b'halo', 0)
        tf0 = timeit(lambda: lib.conv_fwd(x, wf, y, None, stats, N, H, W, Cin, Ho, Wo, Cout, k, k, st, pad, s))
        td0 = timeit(lambda: lib.conv_dgrad(dy, wd, dx, None, N, H, W, Cin, Ho, Wo, Cout, k, k, st, pad, s))
        tw0 = timeit(lambda: lib.conv_wgrad(dy, x, partial0, grad, N, H, W, Cin, Ho, Wo, Cout, k, k, st, pad, nsplit0, pps0, s))
        lib.set_option(b'halo', 1)
        tf = timeit(lambda: lib.conv_fwd(x, wf, y, None, stats, N, H, W, Cin, Ho, Wo, Cout, k, k, st, pad, s))
        td = timeit(lambda: lib.conv_dgrad(dy, wd, dx, None, N, H, W, Cin, Ho, Wo, Cout, k, k, st, pad, s))
        tw = timeit(lambda: lib.conv_wgrad(dy, x, partial, grad, N, H, W, Cin, Ho, Wo, Cout, k, k, st, pad, nsplit, pps, s))
        tot[0] += tf; tot[1] += td; tot[2] += tw
        print(f'{str((N, H, W, Cin, Cout, k, st)):44s} {fl / tf / 1e12:9.1f} {fl / td / 1e12:9.1f} {fl / tw / 1e12:9.1f}   '
              f'{tf * 1e3:.3f} {td * 1e3:.3f} {tw * 1e3:.3f}  nsplit={nsplit}  [halo off: fwd {fl / tf0 / 1e12:.0f} dgrad {fl / td0 / 1e12:.0f} wgrad {fl / tw0 / 1e12:.0f} TF/s]')
    print('sum ms (one call each):', [round(t * 1e3, 3) for t in tot])


if __name__ == '__main__':
    main()
