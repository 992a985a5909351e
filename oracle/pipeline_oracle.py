"""CPU restatement (numpy) of the reference's training image pipeline.  TEST INFRASTRUCTURE ONLY.

* Random decisions - RandomResizedCrop.get_crop_bbox / __call__ and Flip.__call__
  (pipelines/augmentations.py:213-262,263-305 and :640-683) - use only numpy / `random`; they are PINNED
  against boxes and flips captured from the reference itself (tests/golden/pipeline_decisions.npz,
  gen_golden.py).
* Image arithmetic - mmcv.imresize (cv2.resize INTER_LINEAR), mmcv.imflip_ (cv2.flip), mmcv.imnormalize_
  (cv2.subtract / cv2.multiply) - lives in third-party packages that are absent here (mmcv-full 1.2.1,
  opencv-python; `import mmcv` / `import cv2` fail): **parity unpinned**, restated from the published
  OpenCV implementation (modules/imgproc/src/resize.cpp: HResizeLinear / VResizeLinear for 8-bit data,
  INTER_RESIZE_COEF_BITS = 11; core/src/arithm.cpp for a float32 image and float64 scalars).
"""
import random

import numpy as np


def get_crop_bbox(img_shape, area_range, aspect_ratio_range=(3 / 4, 4 / 3), max_attempts=10):
    """augmentations.py:213-262: ten (aspect ratio, area) candidates from numpy's global RNG, the first
    that fits gets its offset from `random.randint`; fallback = centred square."""
    img_h, img_w = img_shape
    area = img_h * img_w
    lo, hi = aspect_ratio_range
    ratios = np.exp(np.random.uniform(np.log(lo), np.log(hi), size=max_attempts))
    areas = np.random.uniform(*area_range, size=max_attempts) * area
    cw = np.round(np.sqrt(areas * ratios)).astype(np.int32)
    ch = np.round(np.sqrt(areas / ratios)).astype(np.int32)
    for i in range(max_attempts):
        if ch[i] <= img_h and cw[i] <= img_w:
            x = random.randint(0, img_w - cw[i])
            y = random.randint(0, img_h - ch[i])
            return x, y, x + cw[i], y + ch[i]
    size = min(img_h, img_w)
    x, y = (img_w - size) // 2, (img_h - size) // 2
    return x, y, x + size, y + size


def sample_crops(num_frames, clip_len, img_shape, area_range, aspect_ratio_range=(3 / 4, 4 / 3),
                 same_on_clip=True, same_across_clip=True):
    """RandomResizedCrop.__call__ (augmentations.py:263-305): one box is drawn before the loop, then a new
    one for every frame that must not share (the configs: same_on_clip = same_across_clip = False)."""
    box = get_crop_bbox(img_shape, area_range, aspect_ratio_range)
    out = []
    for i in range(num_frames):
        is_new_clip = (not same_across_clip) and i % clip_len == 0 and i > 0
        if (not same_on_clip) or is_new_clip:
            box = get_crop_bbox(img_shape, area_range, aspect_ratio_range)
        out.append(box)
    return np.asarray(out, dtype=np.int32)


def sample_flips(num_frames, clip_len, flip_ratio, same_on_clip=True, same_across_clip=True):
    """Flip.__call__ (augmentations.py:640-683): `np.random.rand()` once, then per frame as above."""
    flip = np.random.rand() < flip_ratio
    out = []
    for i in range(num_frames):
        is_new_clip = (not same_across_clip) and i % clip_len == 0 and i > 0
        if (not same_on_clip) or is_new_clip:
            flip = np.random.rand() < flip_ratio
        out.append(bool(flip))
    return np.asarray(out, dtype=np.uint8)


def _coef(n, m, xaxis):
    """cv2 resize: the two source indices and 11-bit weights for every destination index (axis n -> m).
    Columns: an index outside [0, n-1) is clamped and its fraction zeroed (resize.cpp, xofs/alpha loop);
    rows: the indices are clamped when the rows are fetched, the weights stay (yofs/beta loop +
    resizeGeneric_Invoker's clip)."""
    scale = 1.0 / (float(m) / float(n))
    d = np.arange(m, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if xaxis:
        low = s < 0
        f[low] = 0.0
        s[low] = 0
        high = s >= n - 1
        f[high] = 0.0
        s[high] = n - 1
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    w1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return np.clip(s, 0, n - 1), np.clip(s + 1, 0, n - 1), w0, w1


def resize_bilinear_u8(img, out_w, out_h):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC
    (HResizeLinear<uchar,int,short,2048> then VResizeLinear<uchar,int,short,FixedPtCast<.., 22>>)"""
    h, w = img.shape[:2]
    sx, sx1, ax0, ax1 = _coef(w, out_w, True)
    sy, sy1, by0, by1 = _coef(h, out_h, False)
    src = img.astype(np.int64)
    hz = src[:, sx] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None]            # [h][out_w][c]
    r0, r1 = hz[sy], hz[sy1]
    out = (((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def normalize(img_u8, mean, std):
    """mmcv.imnormalize_(float32 copy, mean, std, to_rgb=False)"""
    mean = np.float64(np.asarray(mean).reshape(1, -1))
    stdinv = 1 / np.float64(np.asarray(std).reshape(1, -1))
    x = (img_u8.astype(np.float64) - mean).astype(np.float32)
    return (x.astype(np.float64) * stdinv).astype(np.float32)


def train_pipeline(frames, boxes, flips, out_hw, mean, std, num_clips, clip_len):
    """frames uint8 [B][F][Hs][Ws][3] (F = num_clips*clip_len decoded frames per sample, clip-major) ->
    imgs fp32 [B][num_clips][3][clip_len][H][W]  (FormatShape 'NCTHW', formating.py:270-283)"""
    B, F = frames.shape[:2]
    H, W = out_hw
    out = np.empty((B, num_clips, 3, clip_len, H, W), np.float32)
    for b in range(B):
        for f in range(F):
            left, top, right, bottom = boxes[b * F + f]
            img = resize_bilinear_u8(frames[b, f, top:bottom, left:right], W, H)
            if flips[b * F + f]:
                img = img[:, ::-1]
            out[b, f // clip_len, :, f % clip_len] = normalize(img, mean, std).transpose(2, 0, 1)
    return out
