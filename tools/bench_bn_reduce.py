#!/usr/bin/env python3
"""The statistics reduction of the train step alone (vfs_bn_stats_finalize on the conv kernels' 128-pixel rows), at the row counts of
the ResNet-50 bench batch that still need their own launch (the 16x16 / 8x8 stages finish theirs in the consumer's prologue):
tools/bench_bn_reduce.py [opt=value ...], e.g. bn_reduce_wide=0 for the ticket kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd._lib import get_lib  # noqa: E402

SHAPES = [(2, 1024, 64), (2, 1024, 256), (2, 256, 128), (2, 256, 512), (2, 4096, 64)]   # G, rows per group, C


def main():
    lib = get_lib()
    for kv in [a for a in sys.argv[1:] if '=' in a]:
        k, v = kv.split('=')
        lib.set_option(k.encode(), int(v))
    dev = torch.device('cuda:0')
    s = torch.cuda.current_stream().cuda_stream
    print(' '.join(sys.argv[1:]) or 'defaults')
    for G, bpg, C in SHAPES:
        part = torch.randn(G * bpg, 2, C, device=dev)
        sums = torch.zeros(G, 2, C, dtype=torch.float64, device=dev)
        scratch = torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        bnp, rm, rv = torch.zeros(G, 4, C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
        big = torch.empty(64 << 20, device=dev)      # something else between the launches: the rows do not sit in the reader's L2

        def fn():
            lib.bn_stats_finalize(part, sums, scratch, gamma, beta, bnp, rm, rv, G, bpg, C, float(bpg * 128), 1e-5, 0.1, s)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            big.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f'G {G} rows/group {bpg:5d} C {C:4d}  {part.numel() * 4 / 1e6:5.2f} MB  median {ts[len(ts) // 2]:6.1f} us  min {ts[0]:6.1f} us')


if __name__ == '__main__':
    main()
