#!/usr/bin/env python3
"""Per-layer table of the fp32 evaluation convolutions (csrc/exact_f32.hip) on the DAVIS feature pass: ResNet stem..res4 at 480x854,
N frames per launch.  For every conv_f32 launch: shape, time (HIP events, median of --reps), achieved TFLOP/s against the fp32-input MFMA
peak (157.3) and algorithmic bytes / time against HBM (8 TB/s) - which of the two bounds the layer.
usage (GPU box): python tools/conv_f32_layers.py --model r50 --frames 5 [--reps 10]"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='r50', choices=['r18', 'r50'])
    ap.add_argument('--frames', type=int, default=5)
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    import vfs_amd
    from vfs_amd.engine import shared_engine
    from vfs_amd.exact import exact_state
    from vfs_amd.synthetic import synthetic_weights_
    dev = torch.device('cuda', 0)
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_{args.model}.py'))
    tc = vfs_amd.ConfigDict(cfg.test_cfg)
    bb = dict(cfg.model['backbone'])
    bb['out_indices'], bb['strides'] = tc['out_indices'], tc['strides']          # tools/test.py:129-133
    tracker = vfs_amd.build_model(dict(type='VanillaTracker', backbone=bb), train_cfg=None, test_cfg=tc)
    synthetic_weights_(tracker, seed=5)
    tracker.to(dev).eval()
    eng = shared_engine(dev)
    ex = exact_state(tracker.backbone)
    N, H, W = args.frames, 480, 854
    x4 = torch.randn(N, H, W, 4, device=dev)
    names = []
    conv0 = ex.conv

    def conv(eng_, name, x, n, h, w, relu, res=None, tag=''):
        u = ex.units[name]
        names.append((name, n, h, w, u['cin'], u['cout'], u['k'], u['stride'], u['dil'], res is not None))
        return conv0(eng_, name, x, n, h, w, relu, res=res, tag=tag)
    ex.conv = conv
    times = None
    for rep in range(args.reps + 1):
        del names[:]
        eng.prof = []
        ex.forward(eng, x4, N, H, W, stop_after_out=True)
        torch.cuda.synchronize()
        rows = [(flops, e0.elapsed_time(e1), nbytes) for (kind, flops, e0, e1, nbytes) in eng.prof if kind == 'conv_f32']
        eng.prof = None
        if rep:
            times = np.array([[r[1] for r in rows]]) if times is None else np.vstack([times, [r[1] for r in rows]])
    med = np.median(times, axis=0)
    print(f'{args.model}, {N} frames of {H}x{W}; time = median of {args.reps} (events around each launch)')
    print(f'{"layer":28s} {"in":>16s} {"k/s/d":>7s} {"Cout":>5s} {"tiles":>6s} {"ms":>7s} {"TFLOP/s":>8s} {"of MFMA":>8s} {"GB/s":>7s} {"of HBM":>7s}')
    tot = totf = 0.0
    for (name, n, h, w, cin, cout, k, st, dil, res), (flops, _, nbytes), ms in zip(names, rows, med):
        span = dil * (k - 1) + 1
        pad = ex.units[name]['pad']
        ho, wo = (h + 2 * pad - span) // st + 1, (w + 2 * pad - span) // st + 1
        tiles = -(-n * ho * wo // 128) * -(-cout // 64)
        tf, gb = flops / ms / 1e9, nbytes / ms / 1e6
        print(f'{name:28s} {f"{h}x{w}x{cin}":>16s} {f"{k}/{st}/{dil}":>7s} {cout:5d} {tiles:6d} {ms:7.4f} {tf:8.1f} {tf / 157.3:8.2f} {gb:7.0f} {gb / 8000:7.2f}'
              + (' +res' if res else ''))
        tot += ms; totf += flops
    print(f'total {tot:.3f} ms per {N} frames = {tot / N:.3f} ms/frame, {totf / tot / 1e9:.1f} TFLOP/s')


if __name__ == '__main__':
    main()
