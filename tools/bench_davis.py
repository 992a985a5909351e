#!/usr/bin/env python3
"""DAVIS label-propagation benchmark (BASELINE.json configs[3]): synthetic 480x854 clip, the
reference's test-time settings (res4 features at stride 8, top-10, tau 0.07, 20 preceding frames +
first frame, radius 12 / 18).  Reports ms per propagated frame, the affinity kernel's achieved
TFLOP/s on the ALGORITHMIC (in-mask) work, and label parity against the CPU oracle on the same
bf16 feature bank."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='r18', choices=['r18', 'r50'])
    ap.add_argument('--frames', type=int, default=30)
    ap.add_argument('--parity-frames', type=int, default=3)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'])
    args = ap.parse_args()
    import vfs_amd
    from oracle import vfs_oracle as O
    from vfs_amd.engine import shared_engine
    from vfs_amd.labelprop import extract_features
    depth = 18 if args.model == 'r18' else 50
    dev = torch.device('cuda:0')
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_r{depth}.py'))
    tc = vfs_amd.ConfigDict(cfg.test_cfg)
    tc['precision'] = args.precision
    bb = dict(cfg.model['backbone'])
    bb['out_indices'], bb['strides'] = tc['out_indices'], tc['strides']
    model = vfs_amd.build_model(dict(type='VanillaTracker', backbone=bb), train_cfg=None, test_cfg=tc)
    ref = O.VanillaTracker(depth, dict(tc))
    O.fill_state_dict_(ref, seed=5)
    model.load_state_dict(ref.state_dict(), strict=False)
    model.to(dev).eval()
    T, H, W = args.frames, 480, 854
    g = torch.Generator(device=dev).manual_seed(0)
    # a slowly drifting synthetic clip so that propagation is not pure noise
    base = torch.randn(1, 1, 3, 1, H, W, device=dev, generator=g)
    imgs = base + 0.15 * torch.randn(1, 1, 3, T, H, W, device=dev, generator=g)
    yy, xx = np.mgrid[0:H, 0:W]
    seg = np.zeros((H, W), np.uint8)
    seg[(yy > 100) & (yy < 300) & (xx > 150) & (xx < 400)] = 1
    seg[(yy > 250) & (yy < 420) & (xx > 500) & (xx < 760)] = 2
    seg_t = torch.from_numpy(seg)[None]
    meta = [dict(original_shape=(H, W, 3))]
    eng = shared_engine()

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model(imgs, return_loss=False, ref_seg_map=seg_t, img_meta=meta)
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0
    run()
    out, dt = run()
    # stage timing: features vs propagation
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bank, h, w, C = extract_features(model, eng, imgs.reshape(1, 3, T, H, W), 10, precision=args.precision)
    torch.cuda.synchronize()
    t_feat = time.perf_counter() - t0
    radius = int(tc['neighbor_range']) // 2
    CO = 3
    sbank = torch.rand(T, h * w, CO, device=dev)
    s = eng.stream(dev)
    lpws = torch.empty(96 * h * w * 10 * 2, device=dev)     # vfs_labelprop_workspace_bytes
    f = T - 1
    slots = [0] + list(range(max(0, f - 20), f))
    ks = (ctypes.c_int * len(slots))(*slots)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lp = eng.lib.labelprop_f32 if args.precision == 'fp32' else eng.lib.labelprop
    for _ in range(2):
        lp(bank, sbank, sbank[f], lpws, lpws.numel() * 4, f, ks, len(slots), h, w, C, CO, radius, 0, 10, 0.07, s)
    e0.record()
    for _ in range(5):
        lp(bank, sbank, sbank[f], lpws, lpws.numel() * 4, f, ks, len(slots), h, w, C, CO, radius, 0, 10, 0.07, s)
    e1.record()
    torch.cuda.synchronize()
    t_lp = e0.elapsed_time(e1) / 5 * 1e-3
    mask = O.spatial_neighbor_circle(h, w, 2 * radius)
    alg_flop = 2.0 * float(mask.sum()) * len(slots) * C
    # label parity on the first frames against the oracle fed the same bf16 bank
    P = min(args.parity_frames + 1, T)
    if args.precision == 'fp32':      # the C oracle end to end (features included): must be 0
        from oracle import exact_oracle as X
        lab = X.forward_test(ref.state_dict(), depth, imgs[:, :, :, :P].cpu(), seg, (H, W, 3), tc)
    else:
        feats = bank[:P].float().cpu().permute(2, 0, 1).reshape(1, C, P, h, w)
        lab = O.label_propagate(feats, seg, (H, W), precede_frames=20, topk=10, temperature=0.07,
                                neighbor_range=int(tc['neighbor_range']), with_first=True, normalize=False)
    mism = float((out[0][:P] != lab).mean())
    # J&F of the propagated labels against a "ground truth" that keeps the first-frame masks (the clip is
    # a static scene + noise): exercises the evaluator on the real pipeline; oracle counts on 3 frames
    from vfs_amd import davis_eval as DE
    from oracle import davis_jf as OJ
    gt_clip = np.broadcast_to(seg, (T, H, W)).copy()
    pred_dev = torch.from_numpy(np.ascontiguousarray(out[0])).to(dev)
    gt_dev = torch.from_numpy(gt_clip).to(dev)
    DE.sequence_counts(pred_dev, gt_dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    jf = DE.evaluate_sequences({'synthetic': (pred_dev, gt_dev)})
    torch.cuda.synchronize()
    t_eval = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = OJ.sequence_counts(out[0][:5], gt_clip[:5])
    t_oracle = (time.perf_counter() - t0) / 3
    counts_equal = bool(np.array_equal(DE.sequence_counts(pred_dev[:5], gt_dev[:5]), want))
    res = {'metric': 'DAVIS label propagation', 'model': f'R{depth}', 'frames': T, 'feature_hw': [h, w], 'C': C,
           'ms_per_frame_end_to_end': dt / (T - 1) * 1e3, 'ms_backbone_per_frame': t_feat / T * 1e3,
           'ms_labelprop_kernel_21_key_frames': t_lp * 1e3, 'key_frames': len(slots),
           'precision': args.precision, 'labelprop_algorithmic_TFLOPs': alg_flop / t_lp / 1e12,
           'labelprop_frac_of_mfma_peak': alg_flop / t_lp / (157.3e12 if args.precision == 'fp32' else 2.5e15),
           'label_mismatch_vs_oracle_same_features': mism, 'parity_frames': P - 1,
           'labels_present': sorted(int(v) for v in np.unique(out[0])),
           'jf_eval': {'J&F-Mean': jf['J&F-Mean'], 'J-Mean': jf['J-Mean'], 'F-Mean': jf['F-Mean'],
                       'ms_per_evaluated_frame_gpu': t_eval / (T - 2) * 1e3,
                       'ms_per_evaluated_frame_cpu_oracle': t_oracle * 1e3,
                       'counts_equal_oracle_on_3_frames': counts_equal, 'objects': 2}}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
