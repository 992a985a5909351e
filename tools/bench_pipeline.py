#!/usr/bin/env python3
"""Time the fused crop/resize/flip/normalise kernel on the reference's training shapes
(340x256 decoded frames -> 224x224, 2 frames per sample) and print its HBM roofline fraction."""
import json
import random
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])
from vfs_amd.pipeline import GpuTrainPipeline  # noqa: E402

MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device('cuda:0')
    pipe = GpuTrainPipeline([dict(type='RandomResizedCrop', area_range=(0.2, 1.), same_across_clip=False, same_on_clip=False),
                             dict(type='Resize', scale=(224, 224), keep_ratio=False),
                             dict(type='Flip', flip_ratio=0.5, same_across_clip=False, same_on_clip=False),
                             dict(type='Normalize', mean=MEAN, std=STD, to_bgr=False),
                             dict(type='FormatShape', input_format='NCTHW')], 2, 1)
    np.random.seed(0)
    random.seed(0)
    frames = torch.randint(0, 256, (B, 2, 256, 340, 3), dtype=torch.uint8, device=dev)
    bs, fs = zip(*[pipe.sample(2, (256, 340)) for _ in range(B)])
    boxes, flips = np.concatenate(bs), np.concatenate(fs)
    res = {}
    for name, kw in [('imgs', dict(want_imgs=True, want_x4=False)), ('x4', dict(want_imgs=False, want_x4=True))]:
        for _ in range(3):
            pipe(frames, boxes=boxes, flips=flips, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            pipe(frames, boxes=boxes, flips=flips, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        crop_bytes = float(((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).sum()) * 3
        out_bytes = 2 * B * 224 * 224 * (12 if name == 'imgs' else 8)
        res[name] = dict(ms=round(ms, 4), frames_per_s=round(2 * B / ms * 1e3), GBps=round((crop_bytes + out_bytes) / ms / 1e6, 1),
                         alg_MB=round((crop_bytes + out_bytes) / 1e6, 1))
    print(json.dumps(dict(B=B, **res)))


if __name__ == '__main__':
    main()
