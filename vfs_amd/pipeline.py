"""GPU-side training input pipeline: the reference's
RandomResizedCrop -> Resize -> Flip -> Normalize -> FormatShape('NCTHW') chain (configs/r*_*.py:48-91,
mmaction/datasets/pipelines/augmentations.py, formating.py:222-309) applied to decoded uint8 frames that
are already on the device, by ONE kernel (`vfs_crop_resize_flip_norm`, csrc/pipeline.hip).  Decoding
(DecordInit / SampleFrames / DecordDecode) stays outside.

The random decisions are drawn on the host with the reference's own rules and RNG streams
(`np.random` for the candidates / flips, `random.randint` for the offsets): seeding both reproduces the
reference's boxes and flips (tests/golden/pipeline_decisions.npz)."""
import random

import numpy as np
import torch

from ._lib import get_lib
from .engine import BF16


def get_crop_bbox(img_shape, area_range, aspect_ratio_range=(3 / 4, 4 / 3), max_attempts=10):
    """RandomResizedCrop.get_crop_bbox (augmentations.py:213-262)"""
    assert 0 < area_range[0] <= area_range[1] <= 1
    assert 0 < aspect_ratio_range[0] <= aspect_ratio_range[1]
    img_h, img_w = img_shape
    log_lo, log_hi = np.log(aspect_ratio_range[0]), np.log(aspect_ratio_range[1])
    aspect = np.exp(np.random.uniform(log_lo, log_hi, size=max_attempts))
    target = np.random.uniform(*area_range, size=max_attempts) * (img_h * img_w)
    cand_w = np.round(np.sqrt(target * aspect)).astype(np.int32)
    cand_h = np.round(np.sqrt(target / aspect)).astype(np.int32)
    for w, h in zip(cand_w, cand_h):
        if h <= img_h and w <= img_w:
            x0 = random.randint(0, img_w - w)
            y0 = random.randint(0, img_h - h)
            return x0, y0, x0 + w, y0 + h
    side = min(img_h, img_w)
    x0, y0 = (img_w - side) // 2, (img_h - side) // 2
    return x0, y0, x0 + side, y0 + side


def _new_for_frame(i, clip_len, same_on_clip, same_across_clip):
    return (not same_on_clip) or ((not same_across_clip) and i % clip_len == 0 and i > 0)


def clips_from_pipeline(pipeline_cfg):
    """(num_clips, clip_len) of the frames that reach the augmentations: SampleFrames' values, regrouped by
    Clip2Frame when present (r18 config: 8 clips of 1 frame -> 2 clips of 4 frames, pipelines/loading.py:226-232)"""
    num_clips = clip_len = None
    for step in pipeline_cfg:
        if step['type'] == 'SampleFrames':
            num_clips, clip_len = int(step.get('num_clips', 1)), int(step['clip_len'])
        elif step['type'] == 'Clip2Frame' and num_clips is not None:
            total = num_clips * clip_len
            clip_len = int(step['clip_len'])
            assert total % clip_len == 0
            num_clips = total // clip_len
    if num_clips is None:
        raise ValueError('the pipeline has no SampleFrames step: pass num_clips / clip_len')
    return num_clips, clip_len


class GpuTrainPipeline:
    """Built from the reference's `train_pipeline` list; `__call__(frames)` with frames uint8
    [B][num_clips*clip_len][Hs][Ws][3] on the GPU returns dict(imgs=fp32 [B][num_clips][3][clip_len][H][W])
    (and x4, the bf16 NHWC4 frames, when asked)."""

    def __init__(self, pipeline_cfg, num_clips=None, clip_len=None):
        if num_clips is None or clip_len is None:
            num_clips, clip_len = clips_from_pipeline(pipeline_cfg)
        self.num_clips, self.clip_len = int(num_clips), int(clip_len)
        self.crop = self.flip = None
        self.out_hw, self.mean, self.std = None, None, None
        for step in pipeline_cfg:
            t = step['type']
            if t == 'RandomResizedCrop':
                self.crop = dict(area_range=tuple(step.get('area_range', (0.08, 1.0))),
                                 aspect_ratio_range=tuple(step.get('aspect_ratio_range', (3 / 4, 4 / 3))),
                                 same_on_clip=step.get('same_on_clip', True), same_across_clip=step.get('same_across_clip', True))
            elif t == 'Resize':
                if step.get('keep_ratio', True):
                    raise NotImplementedError('Resize(keep_ratio=True) is not on the GPU pipeline')
                w, h = step['scale']
                self.out_hw = (int(h), int(w))
            elif t == 'Flip':
                if step.get('direction', 'horizontal') != 'horizontal':
                    raise NotImplementedError('vertical Flip')
                self.flip = dict(flip_ratio=float(step.get('flip_ratio', 0.5)), same_on_clip=step.get('same_on_clip', True),
                                 same_across_clip=step.get('same_across_clip', True))
            elif t == 'Normalize':
                if step.get('to_bgr', False):
                    raise NotImplementedError('Normalize(to_bgr=True)')
                self.mean, self.std = [float(v) for v in step['mean']], [float(v) for v in step['std']]
            elif t == 'FormatShape':
                if step.get('input_format') != 'NCTHW':
                    raise NotImplementedError(f"FormatShape {step.get('input_format')}")
            elif t in ('ColorJitter', 'RandomGrayScale', 'RandomGaussianBlur'):
                raise NotImplementedError(f'{t} is not on the GPU pipeline (commented out in the reference configs)')
        if self.out_hw is None or self.mean is None:
            raise ValueError('pipeline needs Resize(scale=..., keep_ratio=False) and Normalize')

    def sample(self, num_frames, img_shape):
        """boxes int32 [F][4], flips uint8 [F] for ONE sample, consuming the RNGs like the reference"""
        Hs, Ws = img_shape
        if self.crop is None:
            boxes = np.tile(np.asarray([[0, 0, Ws, Hs]], np.int32), (num_frames, 1))
        else:
            c = self.crop
            box = get_crop_bbox(img_shape, c['area_range'], c['aspect_ratio_range'])
            rows = []
            for i in range(num_frames):
                if _new_for_frame(i, self.clip_len, c['same_on_clip'], c['same_across_clip']):
                    box = get_crop_bbox(img_shape, c['area_range'], c['aspect_ratio_range'])
                rows.append(box)
            boxes = np.asarray(rows, np.int32)
        if self.flip is None:
            flips = np.zeros(num_frames, np.uint8)
        else:
            f = self.flip
            flip = np.random.rand() < f['flip_ratio']
            vals = []
            for i in range(num_frames):
                if _new_for_frame(i, self.clip_len, f['same_on_clip'], f['same_across_clip']):
                    flip = np.random.rand() < f['flip_ratio']
                vals.append(flip)
            flips = np.asarray(vals, np.uint8)
        return boxes, flips

    def __call__(self, frames, boxes=None, flips=None, want_x4=False, want_imgs=True):
        assert frames.dtype == torch.uint8 and frames.dim() == 5 and frames.shape[-1] == 3, frames.shape
        B, F, Hs, Ws, _ = frames.shape
        assert F == self.num_clips * self.clip_len
        dev = frames.device
        if boxes is None:      # sample by sample: crop boxes of all its frames, then its flips (pipeline order)
            bs, fs = zip(*[self.sample(F, (Hs, Ws)) for _ in range(B)])
            boxes, flips = np.concatenate(bs), np.concatenate(fs)
        H, W = self.out_hw
        Wp = W + (W & 1)
        imgs = torch.empty(B, self.num_clips, 3, self.clip_len, H, W, device=dev) if want_imgs else None
        x4 = torch.empty(self.num_clips * B * self.clip_len, H, Wp, 4, dtype=BF16, device=dev) if want_x4 else None
        bt = torch.as_tensor(np.ascontiguousarray(boxes, dtype=np.int32)).to(dev)
        ft = torch.as_tensor(np.ascontiguousarray(flips, dtype=np.uint8)).to(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == 'cuda' else None
        get_lib().crop_resize_flip_norm(frames.contiguous(), bt, ft, imgs, x4, B, self.num_clips, self.clip_len, Hs, Ws, H, W,
                                        Wp, *self.mean, *self.std, stream)
        out = dict(boxes=boxes, flips=flips)
        if want_imgs:
            out['imgs'] = imgs
        if want_x4:
            out['x4'] = x4
        return out
