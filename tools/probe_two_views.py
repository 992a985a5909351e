#!/usr/bin/env python3
"""Would the deep stages gain from running the two augmented views as two concurrent chains?  A layer-3-like chain of dependent
launches (1x1 1024->256, 3x3 256->256, 1x1 256->1024, each followed by a BatchNorm apply pass) on 64 frames in ONE stream against
the same chain on 2 x 32 frames in TWO streams (the views are independent through the backbone: BatchNorm batches are per view)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfs_amd._lib import get_lib  # noqa: E402


def chain(lib, N, H, W, s, bufs, reps):
    x, w1, w2, w3, y1, y2, y3, st, bnp = bufs[:9]
    M = N * H * W
    for _ in range(reps):
        lib.conv_fwd(x, w1, y1, None, st, N, H, W, 1024, H, W, 256, 1, 1, 1, 0, s)
        lib.bn_act(y1, bnp, None, None, None, y1, M, 256, M, 1, s)
        lib.conv_fwd(y1, w2, y2, None, st, N, H, W, 256, H, W, 256, 3, 3, 1, 1, s)
        lib.bn_act(y2, bnp, None, None, None, y2, M, 256, M, 1, s)
        lib.conv_fwd(y2, w3, y3, None, st, N, H, W, 256, H, W, 1024, 1, 1, 1, 0, s)
        lib.bn_act(y3, bufs[-1], None, None, None, x, M, 1024, M, 1, s)


def make(N, H, W, dev):
    M = N * H * W
    x = torch.randn(N, H, W, 1024, device=dev).to(torch.bfloat16)
    w1 = torch.randn(256, 1, 1, 1024, device=dev).to(torch.bfloat16) * 0.03
    w2 = torch.randn(256, 3, 3, 256, device=dev).to(torch.bfloat16) * 0.02
    w3 = torch.randn(1024, 1, 1, 256, device=dev).to(torch.bfloat16) * 0.06
    y1 = torch.empty(N, H, W, 256, device=dev, dtype=torch.bfloat16)
    y2 = torch.empty(N, H, W, 256, device=dev, dtype=torch.bfloat16)
    y3 = torch.empty(N, H, W, 1024, device=dev, dtype=torch.bfloat16)
    st = torch.empty(((M + 127) // 128) * 2 * 1024 * 2, device=dev)
    bnp = torch.ones(1, 4, 256, device=dev)
    bnp4 = torch.ones(1, 4, 1024, device=dev)
    return [x, w1, w2, w3, y1, y2, y3, st, bnp, bnp4]


def main():
    lib = get_lib()
    dev = torch.device('cuda:0')
    H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = 6
    one = make(64, H, W, dev)
    two = [make(32, H, W, dev), make(32, H, W, dev)]
    s0 = torch.cuda.current_stream()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def run_one():
        chain(lib, 64, H, W, s0.cuda_stream, one[:9] + [one[9]], reps)

    def run_two():
        sa.wait_stream(s0); sb.wait_stream(s0)
        # interleave the host launches so that both streams are fed
        x = [t[:9] + [t[9]] for t in two]
        for r in range(reps):
            for (bufs, s) in ((x[0], sa), (x[1], sb)):
                chain(lib, 32, H, W, s.cuda_stream, bufs, 1)
        s0.wait_stream(sa); s0.wait_stream(sb)

    for name, fn in (('one stream, 64 frames', run_one), ('two streams, 2 x 32 frames', run_two)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f'{name}: {dt * 1e6:8.1f} us per chain of {6 * reps} launches ({dt / (6 * reps) * 1e6:.1f} us per launch)')


if __name__ == '__main__':
    main()
