#!/usr/bin/env python3
"""Golden crop boxes / flip decisions from the REAL reference pipeline classes
(/root/reference/mmaction/datasets/pipelines/augmentations.py: RandomResizedCrop, Flip), run in the build
container only.  The module's import-time dependencies that are absent here (mmcv, skimage, torchvision)
get empty in-memory stand-ins - none of them is on the code path of the two classes except
`mmcv.imflip_`, which is one numpy slice assignment - and the decisions are READ BACK from what the
reference's own `__call__` did to coordinate images (pixel value = its own (row, col)), so every number
stored is produced by the reference's code and RNG use.

Usage: python tests/golden/gen_pipeline_golden.py  (writes tests/golden/pipeline_decisions.npz)"""
import importlib
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import REF, install_mmcv_standin  # noqa: E402


def import_reference_augmentations():
    install_mmcv_standin()
    mmcv = sys.modules['mmcv']

    def imflip_(img, direction='horizontal'):
        assert direction == 'horizontal'
        img[:] = img[:, ::-1].copy()
        return img
    mmcv.imflip_ = imflip_
    mmcv.is_tuple_of = lambda seq, typ: isinstance(seq, tuple) and all(isinstance(v, typ) for v in seq)

    class _Any:
        def __init__(self, *a, **k):
            pass
    for name, attrs in [('skimage', []), ('skimage.util', ['view_as_windows']), ('torchvision', []),
                        ('torchvision.transforms', ['ColorJitter', 'RandomAffine', 'RandomResizedCrop', 'functional'])]:
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, _Any)
        sys.modules[name] = m
    sys.path.insert(0, REF)
    for name in ['mmaction', 'mmaction.datasets', 'mmaction.datasets.pipelines']:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, *name.split('.'))]
        sys.modules[name] = m
    return importlib.import_module('mmaction.datasets.pipelines.augmentations')


def coord_frames(n, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    return [np.stack([yy, xx, np.full_like(yy, i)], -1).astype(np.int32) for i in range(n)]


def main():
    aug = import_reference_augmentations()
    out = {}
    cases = [   # name, (Hs, Ws), num_clips, clip_len, crop kwargs, flip kwargs, samples, seed
        ('r18cfg', (256, 340), 2, 1, dict(area_range=(0.2, 1.), same_across_clip=False, same_on_clip=False),
         dict(flip_ratio=0.5, same_across_clip=False, same_on_clip=False), 24, 7),
        ('clips', (180, 320), 2, 4, dict(area_range=(0.08, 1.), same_across_clip=False, same_on_clip=True),
         dict(flip_ratio=0.5, same_across_clip=False, same_on_clip=True), 12, 11),
        ('shared', (240, 240), 2, 3, dict(area_range=(0.5, 1.)), dict(flip_ratio=0.3), 12, 13),
        ('fallback', (64, 512), 2, 1, dict(area_range=(0.9, 1.), aspect_ratio_range=(0.75, 1.3333333333333333),
                                           same_across_clip=False, same_on_clip=False),
         dict(flip_ratio=0.5, same_across_clip=False, same_on_clip=False), 8, 17),
    ]
    for name, (hs, ws), nclips, clip_len, ck, fk, nsamp, seed in cases:
        np.random.seed(seed)
        random.seed(seed)
        crop, flip = aug.RandomResizedCrop(**ck), aug.Flip(**fk)
        boxes, flips = [], []
        for _ in range(nsamp):
            nf = nclips * clip_len
            res = dict(imgs=coord_frames(nf, hs, ws), img_shape=(hs, ws), clip_len=clip_len, num_clips=nclips,
                       modality='RGB')
            res = crop(res)
            for im in res['imgs']:
                boxes.append([im[0, 0, 1], im[0, 0, 0], im[-1, -1, 1] + 1, im[-1, -1, 0] + 1])
            res['imgs'] = [np.ascontiguousarray(im) for im in res['imgs']]
            res = flip(res)
            for im, b in zip(res['imgs'], boxes[-nf:]):
                flips.append(int(im[0, 0, 1] != b[0]) if b[2] - b[0] > 1 else 0)
        out[name + '/boxes'] = np.asarray(boxes, np.int32)
        out[name + '/flips'] = np.asarray(flips, np.uint8)
        out[name + '/meta'] = np.asarray([hs, ws, nclips, clip_len, nsamp, seed], np.int64)
        out[name + '/area_range'] = np.asarray(ck['area_range'], np.float64)
        out[name + '/flip_ratio'] = np.float64(fk['flip_ratio'])
        out[name + '/same'] = np.asarray([ck.get('same_on_clip', True), ck.get('same_across_clip', True)], np.uint8)
        print(name, out[name + '/boxes'][:3].tolist(), out[name + '/flips'][:8].tolist())
    path = os.environ.get('VFS_GOLDEN_OUT', HERE)
    np.savez_compressed(os.path.join(path, 'pipeline_decisions.npz'), **out)


if __name__ == '__main__':
    main()
