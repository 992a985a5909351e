#!/usr/bin/env python3
"""Label maps of the REAL reference's `VanillaTracker.forward_test` at DAVIS size ("missing" #1).

Runs only in the build container (imports /root/reference through the mmcv stand-in of gen_golden.py).  Every label map stored
below is the output of the reference's own `VanillaTracker.forward_test` (trackers/vanilla_tracker.py:80-206) with the reference's
own `ResNet`, `masked_attention_efficient`, `spatial_neighbor` and `pil_nearest_interpolate` on torch-CPU fp32:

  forward_test_r50_davis.npz   ResNet-50, the shipped test-time config unchanged (strides (1,2,1,1), res4 = 1024 channels,
                               neighbor_range 36 = radius 18, precede_frames 20, top-10, temperature 0.07), 480x854, T frames
                               (feature map 60x107, the first frame doubled in the key set as vanilla_tracker.py:133-149 does)
  forward_test_r18_davis.npz   ResNet-18, shipped config (neighbor_range 24), 480x854
  forward_test_r50_small.npz   ResNet-50, 96x128, 9 frames, precede_frames 3 (the key window slides from frame 4 on), radius 4

Inputs and weights are closed-form fills (oracle/vfs_oracle.py), so the fixtures hold outputs only: the uint8 label maps, their
SHA-256, and samples of the res4 features.  The clip is the drifting scene of tests/test_exact_f32.py (base + per-frame
perturbation) so that propagation carries labels across frames.

Usage:  python tests/golden/gen_davis_golden.py [T_r50 [T_r18]]      (defaults 7 and 7; ~1 minute per R50 frame on 8 cores)
"""
import hashlib
import os
import runpy
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402
from oracle.vfs_oracle import fill_state_dict_, fill_tensor  # noqa: E402

REF = GG.REF
CFG = {18: 'r18_nc_sgd_cos_100e_r2_1xNx8_k400.py', 50: 'r50_nc_sgd_cos_100e_r5_1xNx2_k400.py'}


def davis_clip(T, H=480, W=854):
    """the clip and first-frame labels of test_full_size_davis_clip_fp32_bit_exact_vs_oracle (tests/test_exact_f32.py)"""
    imgs = fill_tensor([1, 1, 3, 1, H, W], 43, scale=2.0) + 0.3 * fill_tensor([1, 1, 3, T, H, W], 44, scale=2.0)
    yy, xx = np.mgrid[0:H, 0:W]
    seg = np.zeros((H, W), np.uint8)
    seg[(yy > 100) & (yy < 300) & (xx > 150) & (xx < 400)] = 1
    seg[(yy > 250) & (yy < 450) & (xx > 500) & (xx < 800)] = 2
    seg[(yy - 120) ** 2 + (xx - 650) ** 2 < 80 ** 2] = 3
    return imgs, seg


def small_clip(T, H=96, W=128):
    imgs = fill_tensor([1, 1, 3, 1, H, W], 45, scale=2.0) + 0.3 * fill_tensor([1, 1, 3, T, H, W], 46, scale=2.0)
    yy, xx = np.mgrid[0:H, 0:W]
    seg = np.zeros((H, W), np.uint8)
    seg[(yy > 20) & (yy < 60) & (xx > 30) & (xx < 70)] = 1
    seg[(yy > 50) & (yy < 90) & (xx > 80) & (xx < 120)] = 2
    return imgs, seg


def run(builder, common, depth, imgs, seg, **override):
    cfg = runpy.run_path(os.path.join(REF, 'configs', CFG[depth]))
    tc = GG.AttrDict(cfg['test_cfg'])
    tc.update(override)
    bb = dict(cfg['model']['backbone'])
    bb['out_indices'], bb['strides'] = tc['out_indices'], tc['strides']       # tools/test.py:129-133
    model = builder.build_model(dict(type='VanillaTracker', backbone=bb), train_cfg=None, test_cfg=tc)
    fill_state_dict_(model, seed=5)
    model.eval()
    H, W = seg.shape
    t0 = time.time()
    with torch.no_grad():
        res = model(imgs, return_loss=False, ref_seg_map=torch.from_numpy(seg)[None], img_meta=[dict(original_shape=(H, W, 3))])
        feat = model.extract_feat_test(common.video2images(imgs.reshape((-1,) + imgs.shape[2:])[:, :, :2]))
    maps = res[0].astype(np.uint8)
    print(f'R{depth} {tuple(imgs.shape)}: {time.time() - t0:.0f} s, labels per frame', [np.bincount(m.ravel(), minlength=4).tolist() for m in maps[-2:]])
    return dict(seg_preds=maps, ref_seg=seg, sha256=np.array(hashlib.sha256(maps.tobytes()).hexdigest()),
                feat_shape=np.array(feat.shape), feat_sample=feat.flatten()[::9973].numpy().copy(),
                test_cfg=np.array(repr({k: tc[k] for k in sorted(tc) if k != 'output_dir'})))


def main():
    t50 = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    t18 = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    torch.manual_seed(0)
    torch.set_num_threads(8)
    builder, trackers, common = GG.import_reference_hot_path()
    OUT = os.environ.get('VFS_GOLDEN_OUT', HERE)
    save = lambda name, **kw: (np.savez_compressed(os.path.join(OUT, name + '.npz'), **kw), print('wrote', name))
    imgs, seg = small_clip(9)
    save('forward_test_r50_small', **run(builder, common, 50, imgs, seg, neighbor_range=8, precede_frames=3))
    imgs, seg = davis_clip(t18)
    save('forward_test_r18_davis', **run(builder, common, 18, imgs, seg))
    imgs, seg = davis_clip(t50)
    save('forward_test_r50_davis', **run(builder, common, 50, imgs, seg))


if __name__ == '__main__':
    main()
