#!/usr/bin/env python3
"""Where does a DAVIS clip's wall time go?  Synchronised phase timings of VanillaTracker.forward_test (fp32 path) on a synthetic
480x854 clip: feature extraction, the propagation loop (GPU time vs host enqueue time), the final device-to-host copy."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import vfs_amd      # noqa: E402
from vfs_amd import labelprop as LP      # noqa: E402
from vfs_amd.synthetic import synthetic_weights_      # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T = int(sys.argv[2]) if len(sys.argv) > 2 else 31
dev = torch.device('cuda:0')
cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_r{depth}.py'))
tc = vfs_amd.ConfigDict(cfg.test_cfg)
bb = dict(cfg.model['backbone'])
bb['out_indices'], bb['strides'] = tc['out_indices'], tc['strides']
model = vfs_amd.build_model(dict(type='VanillaTracker', backbone=bb), train_cfg=None, test_cfg=tc)
synthetic_weights_(model, seed=5)
model.to(dev).eval()
H, W = 480, 854
imgs = torch.randn(1, 1, 3, T, H, W, device=dev)
seg = np.zeros((H, W), np.uint8)
seg[100:300, 150:400] = 1
seg_t, meta = torch.from_numpy(seg)[None], [dict(original_shape=(H, W, 3))]
for _ in range(2):
    model(imgs, return_loss=False, ref_seg_map=seg_t, img_meta=meta)
torch.cuda.synchronize()

orig_extract = LP.extract_features
marks = {}


def timed_extract(*a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = orig_extract(*a, **k)
    marks['extract_enqueue'] = time.perf_counter() - t0
    torch.cuda.synchronize()
    marks['extract_total'] = time.perf_counter() - t0
    marks['t_after_extract'] = time.perf_counter()
    return r


LP.extract_features = timed_extract
orig_cpu = torch.Tensor.cpu


def timed_cpu(self, *a, **k):
    if self.is_cuda and self.numel() > 1 << 20:
        marks['loop_enqueue'] = time.perf_counter() - marks['t_after_extract']
        torch.cuda.synchronize()
        marks['loop_total'] = time.perf_counter() - marks['t_after_extract']
        t0 = time.perf_counter()
        r = orig_cpu(self, *a, **k)
        marks['d2h'] = time.perf_counter() - t0
        return r
    return orig_cpu(self, *a, **k)


torch.Tensor.cpu = timed_cpu
torch.cuda.synchronize()
t0 = time.perf_counter()
model(imgs, return_loss=False, ref_seg_map=seg_t, img_meta=meta)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f'R{depth}, {T} frames: total {tot * 1e3:.1f} ms = {tot / (T - 1) * 1e3:.2f} ms per propagated frame')
for k in ('extract_enqueue', 'extract_total', 'loop_enqueue', 'loop_total', 'd2h'):
    print(f'  {k:16s} {marks[k] * 1e3:8.1f} ms')
print(f'  before extract + after copy (host set-up, PIL resize, numpy): {(tot - marks["extract_total"] - marks["loop_total"] - marks["d2h"]) * 1e3:.1f} ms')
