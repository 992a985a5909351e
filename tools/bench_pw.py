#!/usr/bin/env python3
"""1x1 (pointwise) convolutions of ResNet-50 at the bench batch (64 frames of 256x256): time and algorithmic
GB/s of forward and dgrad through the implicit-GEMM kernel, for A/B of its options
(tools/bench_pw.py [fbn] [opt=value ...], e.g. igemm_onek=0 igemm_bc=64; fbn also times the dgrad with the
mask-gated identity add and the fused BatchNorm-backward statistics)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd._lib import get_lib  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout
    (64, 64, 64, 64, 64), (64, 64, 64, 64, 256), (64, 64, 64, 256, 64), (64, 32, 32, 128, 512), (64, 32, 32, 512, 128),
    (64, 16, 16, 256, 1024), (64, 16, 16, 1024, 256), (64, 8, 8, 512, 2048), (64, 8, 8, 2048, 512),
    (64, 64, 64, 256, 128), (64, 32, 32, 512, 256), (64, 16, 16, 1024, 512),      # conv1 of the first block of layers 2-4 (stride sits in conv2)
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
    lib = get_lib()
    fbn = 'fbn' in sys.argv[1:]
    for kv in [a for a in sys.argv[1:] if '=' in a]:
        k, v = kv.split('=')
        lib.set_option(k.encode(), int(v))
    dev = torch.device('cuda:0')
    s = torch.cuda.current_stream().cuda_stream
    print(' '.join(sys.argv[1:]) or 'defaults')
    tot = [0.0, 0.0, 0.0]
    for (N, H, W, Cin, Cout) in SHAPES:
        M = N * H * W
        x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
        wf = torch.randn(Cout, 1, 1, Cin, device=dev).to(torch.bfloat16)
        wd = torch.randn(Cin, 1, 1, Cout, device=dev).to(torch.bfloat16)
        y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
        dy = torch.randn(N, H, W, Cout, device=dev).to(torch.bfloat16)
        dx = torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
        stats = torch.empty((M + 127) // 128 * 2 * Cout, device=dev)
        nbytes = 2.0 * (M * Cin + M * Cout + Cin * Cout)
        tf = timeit(lambda: lib.conv_fwd(x, wf, y, None, stats, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, s))
        td = timeit(lambda: lib.conv_dgrad(dy, wd, dx, None, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, s))
        tot[0] += tf; tot[1] += td
        line = f'{str((N, H, W, Cin, Cout)):28s} fwd {tf * 1e6:7.1f} us {nbytes / tf / 1e9:7.0f} GB/s   dgrad {td * 1e6:7.1f} us {nbytes / td / 1e9:7.0f} GB/s'
        if fbn and M % 128 == 0:
            # the dgrad as the train step runs it at a block input: + identity gradient gated by the mask bits + BatchNorm-backward
            # statistics of the [M][Cin] tensor it produces (raw x of that unit + its mask bits as operands)
            add = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
            xr = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
            bits = torch.randint(0, 256, (M * Cin // 8,), device=dev, dtype=torch.uint8)
            bnp = torch.rand(1, 4, Cin, device=dev) + 0.5
            part = torch.empty((M // 128) * 2 * Cin, device=dev)
            tb = timeit(lambda: lib.conv_dgrad_bn_maskadd(dy, wd, dx, add, bits, xr, bits, bnp, part, M, 2, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, s))
            fb = 2.0 * (M * Cout + 3 * M * Cin + Cin * Cout) + 2.0 * M * Cin / 8
            tot[2] += tb
            line += f'   dgrad+add+stats {tb * 1e6:7.1f} us {fb / tb / 1e9:7.0f} GB/s'
        print(line)
    print(f'sum us: fwd {tot[0] * 1e6:.1f} dgrad {tot[1] * 1e6:.1f}' + (f' dgrad+add+stats {tot[2] * 1e6:.1f}' if fbn else ''))


if __name__ == '__main__':
    main()
