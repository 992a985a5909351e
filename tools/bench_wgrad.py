#!/usr/bin/env python3
"""1x1 weight gradients of ResNet-50 at the bench batch (64 frames of 256x256): the generic split-K kernel alone (reduction
deferred: grad = NULL) and with its reduction, per split plan.  tools/bench_wgrad.py [target_blocks ...] [option=value ...] (default 256);
VFS_HIP_LIB=<path> times a variant build; VFS_WGRAD_ONLY=<index> runs one shape only (for PMC passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd._lib import VfsLib, get_lib, set_lib  # noqa: E402
from vfs_amd.packing import wgrad_splits  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout
    (64, 64, 64, 64, 64), (64, 64, 64, 64, 256), (64, 64, 64, 256, 64), (64, 32, 32, 128, 512), (64, 32, 32, 512, 128),
    (64, 16, 16, 256, 1024), (64, 16, 16, 1024, 256), (64, 8, 8, 512, 2048), (64, 8, 8, 2048, 512),
]


def timeit(fn, iters=30):
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
    if os.environ.get('VFS_HIP_LIB'):
        set_lib(VfsLib(os.environ['VFS_HIP_LIB']))
    lib = get_lib()
    dev = torch.device('cuda:0')
    s = torch.cuda.current_stream().cuda_stream
    only = os.environ.get('VFS_WGRAD_ONLY')
    for kv in [a for a in sys.argv[1:] if '=' in a]:      # library options: name=value
        k, v = kv.split('=')
        lib.set_option(k.encode(), int(v))
    plans = [int(a) for a in sys.argv[1:] if '=' not in a] or [256]
    for tb in plans:
        tot = 0.0
        for idx, (N, H, W, Cin, Cout) in enumerate(SHAPES):
            if only is not None and idx != int(only):
                continue
            M = N * H * W
            x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
            dy = torch.randn(N, H, W, Cout, device=dev).to(torch.bfloat16)
            nsplit, pps = wgrad_splits(M, Cout, Cin, target_blocks=tb)
            partial = torch.empty(nsplit * Cout * Cin, device=dev)
            grad = torch.zeros(Cout, Cin, 1, 1, device=dev)
            tk = timeit(lambda: lib.conv_wgrad(dy, x, partial, None, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, nsplit, pps, s))
            tr = timeit(lambda: lib.conv_wgrad(dy, x, partial, grad, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, nsplit, pps, s))
            tot += tr
            fl, by = 2.0 * M * Cin * Cout, 2.0 * M * (Cin + Cout)
            print(f'tb {tb:4d} {str((N, H, W, Cin, Cout)):26s} nsplit {nsplit:4d} steps/split {pps // 64:3d}  kernel {tk * 1e6:6.1f} us '
                  f'({fl / tk / 1e12:5.0f} TF/s, {by / tk / 1e9:5.0f} GB/s)  with reduce {tr * 1e6:6.1f} us')
        print(f'tb {tb}: sum with reduce {tot * 1e6:.1f} us')


if __name__ == '__main__':
    main()
