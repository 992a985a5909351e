#!/usr/bin/env python3
"""How far does the LOSS of one bf16-storage emulation of the train step land from the fp32 oracle's, over many draws?  (advisor
finding r04: the HIP step's loss error, 9.4e-4 on ResNet-18 [32,2,3,1,224,224], sat ~2x outside the four draws of
tests/test_cfg1_golden.py, 6e-5 .. 5e-4.)  The loss is ONE scalar per draw: four draws cannot bound its scatter.  Here: the three
statistics variants x jittered inputs (every frame value moved by one fp32 ulp, seeds 0..N-1).
usage: tools/parity_loss_scatter.py [depth] [draws]   ->  profiles/r05_parity_loss_scatter_r<depth>.json"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import vfs_oracle as O  # noqa: E402
from tests.test_cfg1_golden import _jitter, _oracle_step  # noqa: E402


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    ndraw = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    imgs = O.fill_tensor([32, 2, 3, 1, 224, 224], seed=11, scale=2.0)
    _, _, log32, _ = _oracle_step(depth, imgs, False)
    ref = float(log32['loss'])
    errs = []
    for k in range(ndraw):
        stats = ('stored', 'engine', 'acc')[k % 3]
        x = _jitter(imgs, k) if k >= 3 else imgs
        _, _, log, _ = _oracle_step(depth, x, True, stats=stats)
        errs.append((stats, k if k >= 3 else None, abs(float(log['loss']) - ref)))
        print(errs[-1], flush=True)
    v = sorted(e for _, _, e in errs)
    out = {'depth': depth, 'loss_fp32': ref, 'draws': errs, 'min': v[0], 'median': v[len(v) // 2], 'max': v[-1]}
    json.dump(out, open(os.path.join(REPO, 'profiles', f'r05_parity_loss_scatter_r{depth}.json'), 'w'), indent=1)
    print(out)


if __name__ == '__main__':
    main()
