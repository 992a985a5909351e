"""VanillaTracker.forward_test on the HIP kernels (trackers/vanilla_tracker.py:80-206).

Differences from the reference's execution (not its results): the feature bank and the soft
label bank stay on the GPU for the whole clip (the reference keeps them on the CPU and re-uploads
<= 21 frames per step), features are L2-normalised once per frame, the backbone stops after the
evaluated stage (the reference also computes and discards layer4), and the dense [T*HW, HW]
affinity / boolean mask are never materialised."""
import ctypes

import numpy as np
import torch

from .engine import BF16, shared_engine


def pil_nearest_resize(label, out_h, out_w):
    """PIL NEAREST as used by mmcv.imresize(backend='pillow') in pil_nearest_interpolate
    (common/utils.py:25-42): src = floor((dst + 0.5) * in / out)."""
    in_h, in_w = label.shape
    ys = np.minimum(np.floor((np.arange(out_h) + 0.5) * (in_h / out_h)).astype(np.int64), in_h - 1)
    xs = np.minimum(np.floor((np.arange(out_w) + 0.5) * (in_w / out_w)).astype(np.int64), in_w - 1)
    return label[ys][:, xs]


def torch_nearest_resize(label, out_h, out_w):
    """F.interpolate(mode='nearest') index rule: src = floor(dst * in / out)."""
    in_h, in_w = label.shape
    ys = np.minimum(np.floor(np.arange(out_h) * np.float32(in_h / out_h)).astype(np.int64), in_h - 1)
    xs = np.minimum(np.floor(np.arange(out_w) * np.float32(in_w / out_w)).astype(np.int64), in_w - 1)
    return label[ys][:, xs]


def extract_features(tracker, eng, frames_ncthw, batch_step, all_blocks=False):
    """imgs [1,3,T,H,W] fp32 -> L2-normalised bf16 bank [T, h*w, C] of the evaluated stage; with
    all_blocks (vanilla_tracker.py:32-45, README.md:76) a LIST of banks, one per residual block of
    every stage in test_cfg.out_indices."""
    bb = tracker.backbone
    dev = frames_ncthw.device
    _, _, T, H, W = frames_ncthw.shape
    bb.attach(eng)
    eng.pack_weights()
    Wp = W + (W & 1)
    s = eng.stream(dev)
    stages = tuple(tracker.test_cfg.get('out_indices', bb.out_indices))
    stage = stages[0]
    banks, shapes = None, None
    for t0 in range(0, T, batch_step):
        n = min(batch_step, T - t0)
        x4 = eng.buf('backbone.x4', (n, H, Wp, 4), BF16, dev)
        chunk = frames_ncthw[:, :, t0:t0 + n].contiguous().float()
        eng.lib.imgs_to_nhwc4(chunk, x4, 1, 1, n, H, W, Wp, s)
        outs, ctx = bb.forward_nhwc(eng, x4, n, H, W, 1, False, stop_after_out=True)
        if all_blocks:          # every block output of the listed stages, in network order
            feats, bi = [], 0
            for si, lname in enumerate(bb.res_layers):
                nb = len(getattr(bb, lname))
                if si in stages:
                    for b in ctx['blocks'][bi:bi + nb]:
                        feats.append((b['out'], b['dims'][-1][2], b['dims'][-1][3], b['out'].shape[-1]))
                bi += nb
                if bi >= len(ctx['blocks']):
                    break
        else:
            feats = [outs[stage]]
        if banks is None:
            banks = [torch.empty(T, h * w, C, dtype=BF16, device=dev) for (_, h, w, C) in feats]
            shapes = [(h, w, C) for (_, h, w, C) in feats]
        for bank, (feat, h, w, C) in zip(banks, feats):
            eng.lib.l2norm_rows(feat, bank[t0:t0 + n], n * h * w, C, s)
    if all_blocks:
        return banks, shapes
    return banks[0], shapes[0][0], shapes[0][1], shapes[0][2]


def forward_test_hip(tracker, imgs, ref_seg_map, img_meta):
    tc = tracker.test_cfg
    eng = shared_engine()
    imgs = imgs.reshape((-1,) + tuple(imgs.shape[2:]))          # [1,3,T,H,W]
    assert imgs.shape[0] == 1
    dev = imgs.device
    clip_len = imgs.size(2)
    if tracker.training:
        raise RuntimeError('forward_test expects model.eval() (BatchNorm running statistics)')
    if ref_seg_map.ndim == 4:
        raise NotImplementedError('one-hot reference maps are not on the HIP path yet')
    nr = tc.get('neighbor_range', None)
    radius = int(nr) // 2 if nr is not None else 0
    if tc.get('with_first_neighbor', True) is False:
        raise NotImplementedError('with_first_neighbor=False (unmasked first frame) is not on the HIP path yet')
    if not tc.get('with_norm', True):
        raise NotImplementedError('with_norm=False')
    all_blocks = bool(tc.get('all_blocks', False))
    if all_blocks:
        banks, shapes = extract_features(tracker, eng, imgs, int(tc.get('batch_step', 10)), all_blocks=True)
    else:
        bank, h, w, C = extract_features(tracker, eng, imgs, int(tc.get('batch_step', 10)))
        banks, shapes = [bank], [(h, w, C)]
    s = eng.stream(dev)
    out_h, out_w = img_meta[0]['original_shape'][:2]
    ref = ref_seg_map[0].detach().cpu().numpy().astype(np.uint8)
    precede = int(tc['precede_frames'])
    topk, temp = int(tc['topk']), float(tc['temperature'])
    all_preds = []
    for bank, (h, w, C) in zip(banks, shapes):
        small = pil_nearest_resize(ref, h, w)
        CO = int(small.max()) + 1                                    # F.one_hot infers max+1 classes
        sbank = torch.zeros(clip_len, h * w, CO, dtype=torch.float32, device=dev)
        eng.lib.onehot(torch.from_numpy(np.ascontiguousarray(small)).to(dev), sbank[0], h * w, CO, s)
        preds = torch.empty(clip_len, out_h, out_w, dtype=torch.uint8, device=dev)
        preds[0] = torch.from_numpy(np.ascontiguousarray(torch_nearest_resize(ref, out_h, out_w))).to(dev)
        partial = eng.ws('ws.segpost', 64 * CO * 2, torch.float32, dev)
        lpws = eng.ws('ws.labelprop', 24 * h * w * 10 * 2, torch.float32, dev)
        for f in range(1, clip_len):
            key_start = max(0, f - precede)
            slots = list(range(key_start, f))
            if tc.get('with_first', True):
                slots = [0] + slots                                 # frame 0 twice while f <= precede (as the reference)
            ks = (ctypes.c_int * len(slots))(*slots)
            eng.lib.labelprop(bank, sbank, sbank[f], lpws, f, ks, len(slots), h, w, C, CO, radius, topk, temp, s)
            eng.lib.seg_postprocess(sbank[f], partial, preds[f], h, w, CO, out_h, out_w, s)
        all_preds.append(preds.cpu().numpy())
    if len(all_preds) > 1:      # vanilla_tracker.py:199-205: [1, num_feats, T, H, W] unravelled over the batch dim
        return [np.stack(all_preds, axis=0)]
    return [all_preds[0]]
