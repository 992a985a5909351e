"""VanillaTracker.forward_test on the HIP kernels (trackers/vanilla_tracker.py:80-206).

Differences from the reference's execution (not its results): the feature bank and the soft
label bank stay on the GPU for the whole clip (the reference keeps them on the CPU and re-uploads
<= 21 frames per step), features are L2-normalised once per frame, the backbone stops after the
evaluated stage (the reference also computes and discards layer4), and the dense [T*HW, HW]
affinity / boolean mask are never materialised.

Two precisions (test_cfg.precision, default 'fp32'; env VFS_EVAL_PRECISION overrides the default):
  'fp32'  fp32 storage and bit-defined fp32 arithmetic (csrc/exact_f32.hip, vfs_amd/exact.py) - what the
          reference computes in; label maps equal the C oracle's bit for bit;
  'bf16'  the training path's bf16 kernels and a bf16 bank (about 3x faster; labels differ where the exact
          10th / 11th affinities or the two best classes are nearly tied).
Limits of the kernels (checked, errors raised): topk <= 10, at most 64 key frames per step
(precede_frames + 1 <= 64), feature channels % 64 == 0 (bf16) / % 4 == 0 (fp32), <= 256 classes."""
import ctypes
import os
import tempfile

import numpy as np
import torch

from .engine import BF16, shared_engine

F32 = torch.float32


def eval_precision(test_cfg):
    p = None if test_cfg is None else test_cfg.get('precision', None)
    p = p or os.environ.get('VFS_EVAL_PRECISION', 'fp32')
    if p not in ('fp32', 'bf16'):
        raise ValueError(f"test_cfg.precision must be 'fp32' or 'bf16', got {p!r}")
    return p


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


def mask_pairs(h, w, radius):
    """number of (query, key) pairs of one key frame inside the circular mask (affinity_utils.py:144-156: distance < radius);
    radius <= 0: no mask -> (h*w)^2.  The ALGORITHMIC affinity work of a propagation step is 2 * C * pairs per key frame."""
    if radius <= 0:
        return float(h * w) ** 2
    n = 0
    for dy in range(-(radius - 1), radius):
        for dx in range(-(radius - 1), radius):
            if dy * dy + dx * dx < radius * radius:
                n += max(0, h - abs(dy)) * max(0, w - abs(dx))
    return float(n)


def _block_feats(bb, ctx_blocks, stages):
    """(block index range) of every residual block of the listed stages, in network order"""
    sel, bi = [], 0
    for si, lname in enumerate(bb.res_layers):
        nb = len(getattr(bb, lname))
        if si in stages:
            sel += list(range(bi, min(bi + nb, ctx_blocks)))
        bi += nb
        if bi >= ctx_blocks:
            break
    return sel


def two_pass_ok(precision, with_norm, C):
    """the two-pass exact label propagation (csrc/labelprop2.hip: bf16 hi/lo prefilter + exact rescoring, same bits as the dense
    kernel) needs the fp32 path, unit rows and a channel count its register-resident query tile covers; VFS_LP_TWO_PASS=0 keeps
    the dense kernel (A/B)"""
    return precision == 'fp32' and with_norm and C in (256, 512, 1024) and os.environ.get('VFS_LP_TWO_PASS', '1') == '1'


def extract_features(tracker, eng, frames_ncthw, batch_step, all_blocks=False, precision='bf16', with_norm=True, split_banks=None):
    """imgs [1,3,T,H,W] fp32 -> feature bank [T, h*w, C] of the evaluated stage, L2-normalised over the
    channels unless with_norm=False (bf16 or fp32 by `precision`); with all_blocks (vanilla_tracker.py:32-45,
    README.md:76) a LIST of banks, one per residual block of every stage in test_cfg.out_indices.
    split_banks: a list that receives, per bank, its bf16 hi / lo split copy [T, h*w, 2C] (or None) for the two-pass kernels."""
    bb = tracker.backbone
    dev = frames_ncthw.device
    _, _, T, H, W = frames_ncthw.shape
    s = eng.stream(dev)
    stages = tuple(tracker.test_cfg.get('out_indices', bb.out_indices))
    stage = stages[0]
    exact = precision == 'fp32'
    if exact:
        from .exact import exact_state
        ex = exact_state(bb)
    else:
        bb.attach(eng)
        eng.pack_weights()
    Wp = W + (W & 1)
    banks, shapes = None, None
    for t0 in range(0, T, batch_step):
        n = min(batch_step, T - t0)
        chunk = frames_ncthw[:, :, t0:t0 + n].contiguous().float()
        if exact:
            x4 = eng.buf('exact.x4', (n, H, W, 4), F32, dev)
            eng.lib.imgs_to_nhwc4_f32(chunk, x4, 1, 1, n, H, W, s)
            outs, blocks = ex.forward(eng, x4, n, H, W, stop_after_out=True)
            block_feats = [b[:4] for b in blocks]
        else:
            x4 = eng.buf('backbone.x4', (n, H, Wp, 4), BF16, dev)
            eng.lib.imgs_to_nhwc4(chunk, x4, 1, 1, n, H, W, Wp, s)
            outs, ctx = bb.forward_nhwc(eng, x4, n, H, W, 1, False, stop_after_out=True)
            block_feats = [(b['out'], b['dims'][-1][2], b['dims'][-1][3], b['out'].shape[-1]) for b in ctx['blocks']]
        if all_blocks:          # every block output of the listed stages, in network order
            feats = [block_feats[i] for i in _block_feats(bb, len(block_feats), stages)]
        else:
            feats = [outs[stage]]
        if banks is None:
            banks = [torch.empty(T, h * w, C, dtype=F32 if exact else BF16, device=dev) for (_, h, w, C) in feats]
            shapes = [(h, w, C) for (_, h, w, C) in feats]
            hls = [torch.empty(T, h * w, 2 * C, dtype=BF16, device=dev) if two_pass_ok(precision, with_norm, C) else None
                   for (_, h, w, C) in feats]
            if split_banks is not None:
                split_banks.extend(hls)
        for bank, hl, (feat, h, w, C) in zip(banks, hls, feats):
            if not with_norm:
                bank[t0:t0 + n].copy_(feat.reshape(n, h * w, C))
            elif exact:
                eng.lib.l2norm_rows_f32(feat, bank[t0:t0 + n], n * h * w, C, s)
                if hl is not None:      # x = hi + lo (bf16 each) of the unit rows: the operands of the two-pass kernel's matrix pass
                    eng.lib.split_rows_bf16x2(bank[t0:t0 + n], hl[t0:t0 + n], n * h * w, C, s)
            else:
                eng.lib.l2norm_rows(feat, bank[t0:t0 + n], n * h * w, C, s)
    if all_blocks:
        return banks, shapes
    return banks[0], shapes[0][0], shapes[0][1], shapes[0][2]


def forward_test_hip(tracker, imgs, ref_seg_map, img_meta):
    tc = tracker.test_cfg
    eng = shared_engine()
    precision = eval_precision(tc)
    exact = precision == 'fp32'
    imgs = imgs.reshape((-1,) + tuple(imgs.shape[2:]))          # [1,3,T,H,W]
    assert imgs.shape[0] == 1
    dev = imgs.device
    clip_len = imgs.size(2)
    if tracker.training:
        raise RuntimeError('forward_test expects model.eval() (BatchNorm running statistics)')
    nr = tc.get('neighbor_range', None)
    radius = int(nr) // 2 if nr is not None else 0
    with_first = bool(tc.get('with_first', True))
    non_mask_len = 0 if tc.get('with_first_neighbor', True) else 1      # vanilla_tracker.py:158-159
    with_norm = bool(tc.get('with_norm', True))
    all_blocks = bool(tc.get('all_blocks', False))
    precede = int(tc['precede_frames'])
    topk, temp = int(tc['topk']), float(tc['temperature'])
    if topk > 10 or precede + (1 if with_first else 0) > 64:
        raise NotImplementedError(f'label propagation kernels: topk <= 10 (got {topk}), precede_frames + first frame <= 64 '
                                  f'(got {precede + (1 if with_first else 0)})')
    hls = []
    if all_blocks:
        banks, shapes = extract_features(tracker, eng, imgs, int(tc.get('batch_step', 10)), True, precision, with_norm, hls)
    else:
        bank, h, w, C = extract_features(tracker, eng, imgs, int(tc.get('batch_step', 10)), False, precision, with_norm, hls)
        banks, shapes = [bank], [(h, w, C)]
    s = eng.stream(dev)
    out_h, out_w = img_meta[0]['original_shape'][:2]
    input_onehot = ref_seg_map.ndim == 4                         # vanilla_tracker.py:94
    if input_onehot:
        ref = ref_seg_map[0].detach().to(dev, F32).contiguous()  # [CO, Hr, Wr] soft / one-hot map
    else:
        ref = ref_seg_map[0].detach().cpu().numpy().astype(np.uint8)
    lp = eng.lib.labelprop_f32 if exact else eng.lib.labelprop
    post = eng.lib.seg_postprocess_exact if exact else eng.lib.seg_postprocess
    all_preds = []
    for bank, hl, (h, w, C) in zip(banks, hls, shapes):
        if input_onehot:
            # bilinear to the feature size (values) and to the original size (frame 0 of the output); the soft maps of
            # the later frames are returned as they are, without min-max / argmax (vanilla_tracker.py:101-111,167)
            CO, hr, wr = ref.shape
            sbank = torch.zeros(clip_len, h * w, CO, dtype=F32, device=dev)
            eng.lib.bilinear_resize_f32(ref, sbank[0], CO, hr, wr, h, w, 0, 1, s)
            preds = torch.empty(clip_len, CO, out_h, out_w, dtype=F32, device=dev)
            eng.lib.bilinear_resize_f32(ref, preds[0], CO, hr, wr, out_h, out_w, 0, 0, s)
            ref = preds[0].clone()            # the reference keeps the resized map for the next feature level
        else:
            small = pil_nearest_resize(ref, h, w)
            CO = int(small.max()) + 1                                    # F.one_hot infers max+1 classes
            sbank = torch.zeros(clip_len, h * w, CO, dtype=F32, device=dev)
            eng.lib.onehot(torch.from_numpy(np.ascontiguousarray(small)).to(dev), sbank[0], h * w, CO, s)
            preds = torch.empty(clip_len, out_h, out_w, dtype=torch.uint8, device=dev)
            ref = np.ascontiguousarray(torch_nearest_resize(ref, out_h, out_w))    # vanilla_tracker.py:101-104 (kept, as there)
            preds[0] = torch.from_numpy(ref).to(dev)
        if CO > 256:
            raise NotImplementedError('label propagation kernels: at most 256 classes')
        pairs = mask_pairs(h, w, radius)
        partial = eng.ws('ws.segpost', 64 * CO * 2, F32, dev)
        nbytes = torch.zeros(1, dtype=torch.int64)
        entries = int(tc.get('lp2_entries', os.environ.get('VFS_LP2_ENTRIES', 0)))      # list entries per query of the two-pass kernels (0: full capacity, 237 MB at 60 x 107)
        if hl is None:
            eng.lib.labelprop_workspace_bytes(h, w, nbytes)
        elif entries > 0:
            eng.lib.labelprop_f32_2pass_workspace_bytes_for(h, w, entries, nbytes)
        else:
            eng.lib.labelprop_f32_2pass_workspace_bytes(h, w, nbytes)
        lpws = eng.ws('ws.labelprop', (int(nbytes.item()) + 3) // 4, F32, dev)
        for f in range(1, clip_len):
            key_start = max(0, f - precede)
            slots = list(range(key_start, f))
            if with_first:
                slots = [0] + slots                                 # frame 0 twice while f <= precede (as the reference)
            assert 0 <= non_mask_len < len(slots)                   # local_attention.py:272
            ks = (ctypes.c_int * len(slots))(*slots)
            nmask = len(slots) - non_mask_len
            work = (2.0 * C * (nmask * pairs + non_mask_len * float(h * w) ** 2),          # in-mask affinity FLOP
                    float(bank.element_size()) * (len(set(slots)) + 1) * h * w * C)         # every key / query row once
            if hl is not None:      # same label maps, bit for bit (csrc/labelprop2.hip); algorithmic work as SURVEY section 8(d) counts it:
                # ONE product per in-mask (query, key, channel) (the kernel EXECUTES three bf16 products for it - hi.hi + hi.lo + lo.hi:
                # bench.py reports that as executed_frac) and every key / query row once in both copies (fp32 + split)
                eng.timed('labelprop_2pass', (work[0], 2.0 * work[1]), dev, eng.lib.labelprop_f32_2pass, bank, hl, sbank, sbank[f], lpws, lpws.numel() * 4, f, ks,
                          len(slots), h, w, C, CO, radius, non_mask_len, topk, temp, 1, s)
            else:
                eng.timed('labelprop_f32' if exact else 'labelprop', work, dev, lp, bank, sbank, sbank[f], lpws, lpws.numel() * 4, f, ks, len(slots), h, w,
                          C, CO, radius, non_mask_len, topk, temp, s)
            if input_onehot:
                eng.lib.bilinear_resize_f32(sbank[f], preds[f], CO, h, w, out_h, out_w, 1, 0, s)
            else:
                eng.timed('seg_postprocess', (0.0, 4.0 * h * w * CO + float(out_h * out_w)), dev, post, sbank[f], partial, preds[f],
                          h, w, CO, out_h, out_w, s)
        arr = preds.cpu().numpy()
        if tracker.save_np:                                         # vanilla_tracker.py:184-193
            os.makedirs('.eval', exist_ok=True)
            tmp = tempfile.NamedTemporaryFile(dir='.eval', suffix='.npy', delete=False)
            tmp.close()
            np.save(tmp.name, arr)
            all_preds.append(tmp.name)
        else:
            all_preds.append(arr)
    if tracker.save_np:
        return [all_preds] if len(all_preds) > 1 else [all_preds[0]]
    if len(all_preds) > 1:      # vanilla_tracker.py:199-205: [1, num_feats, T, ...] unravelled over the batch dim
        return [np.stack(all_preds, axis=0)]
    return [all_preds[0]]
