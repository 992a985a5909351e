#!/usr/bin/env python3
"""Micro-benchmark / PMC target: only the 3x3 stride-1 convs (fwd, dgrad, wgrad) of ResNet-18 (default) or ResNet-50 (`r50`)
at bench size.  tools/bench_halo.py [iters] [fdw] [r50] [opt=value ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd._lib import VfsLib, get_lib, set_lib  # noqa: E402
from vfs_amd.packing import wgrad_halo_eligible, wgrad_splits  # noqa: E402

SHAPES = [(256, 64, 64, 64), (256, 32, 32, 128), (256, 16, 16, 256), (256, 8, 8, 512)]


def main():
    if os.environ.get('VFS_HIP_LIB'):      # A/B a variant build of the library
        set_lib(VfsLib(os.environ['VFS_HIP_LIB']))
    lib = get_lib()
    for opt in ('halo',):
        if os.environ.get('VFS_OPT_' + opt.upper()):
            lib.set_option(opt.encode(), int(os.environ['VFS_OPT_' + opt.upper()]))
    dev = torch.device('cuda:0')
    s = torch.cuda.current_stream().cuda_stream
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    what = sys.argv[2] if len(sys.argv) > 2 else 'fdw'
    for kv in [a for a in sys.argv[3:] if '=' in a]:
        k, v = kv.split('=')
        lib.set_option(k.encode(), int(v))
    shapes = [(64, 64, 64, 64), (64, 32, 32, 128), (64, 16, 16, 256), (64, 8, 8, 512)] if 'r50' in sys.argv[3:] else SHAPES
    print(' '.join(sys.argv[1:]))
    for (N, H, W, C) in shapes:
        M = N * H * W
        x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
        wf = torch.randn(C, 3, 3, C, device=dev).to(torch.bfloat16)
        y = torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)
        stats = torch.empty((M + 127) // 128 * 2 * C, device=dev)
        nsplit, pps = wgrad_splits(M, C, 9 * C, halo_geom=(N, H, W, C))
        partial = torch.empty(nsplit * C * 9 * C, device=dev)
        grad = torch.zeros(C, C, 3, 3, device=dev)
        fl = 2.0 * M * C * 9 * C
        fns = {'f': lambda: lib.conv_fwd(x, wf, y, None, stats, N, H, W, C, H, W, C, 3, 3, 1, 1, s),
               'd': lambda: lib.conv_dgrad(x, wf, y, None, N, H, W, C, H, W, C, 3, 3, 1, 1, s),
               'w': lambda: lib.conv_wgrad(y, x, partial, grad, N, H, W, C, H, W, C, 3, 3, 1, 1, nsplit, pps, s)}
        out = []
        for k in what:
            fn = fns[k]
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / iters * 1e-3
            out.append(f'{k} {t * 1e6:7.1f} us {fl / t / 1e12:6.0f} TF/s')
        print((N, H, W, C), ' | '.join(out), flush=True)


if __name__ == '__main__':
    main()
