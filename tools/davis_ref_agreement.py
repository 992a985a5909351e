#!/usr/bin/env python3
"""Where do the label maps of the bit-defined fp32 evaluation path (oracle/exact_oracle.c == the HIP kernels, bit for bit) differ from
the REAL reference's `VanillaTracker.forward_test` at DAVIS size, and why?  (VERDICT r05 "missing" #1 / "weak" #2.)

Build container only: imports /root/reference through the mmcv stand-in of tests/golden/gen_golden.py.  The reference runs
UNCHANGED; its `masked_attention_efficient` is wrapped (not replaced) so that every propagation step's own inputs (query / key
features, value maps) and output are seen.  Per propagated frame f:

  free-running      the reference's label map (= tests/golden/forward_test_r*_davis.npz) against the oracle's own run
  teacher-forced    the oracle's arithmetic (its OWN conv features of the same frames, its k-ordered fp32 chains, its top-10 order)
                    on the reference's value maps of this step: differences made by THIS step alone
  classification    a label can only flip where the reference's own float64 argmax margin m(p) (top-2 classes of the normalised,
                    upsampled scores) is smaller than what the two runs' soft labels differ by around p: every free-running pixel
                    mismatch must satisfy  m(p) <= 4 (d_in(p) + g_in) / r_min + 1e-6  (d_in = largest soft-label difference of this
                    frame between the reference and the oracle's free run over the four feature positions p interpolates; g_in =
                    how far the class channels' extremes over the whole map differ between the two runs - the min-max
                    normalisation couples every pixel to them; r_min = smallest max - min of a class channel, the normalisation's
                    divisor) - otherwise it is UNEXPLAINED (must be 0).  Its cause is then one of
      topk-tie   p interpolates a query whose teacher-forced output differs by more than QDIFF: the top-10 membership of that query
                 changed in THIS step; every such query's exact (float64, on the reference's features) 10th and 11th affinities
                 lie closer than TOL_AFF (checked for all of them: gaps_over_tol must be 0)
      rounding   no such query, but this step's own arithmetic moved the soft labels around p by at least half of d_in
                 (summation order inside softmax / value sum / features, below QDIFF)
      propagated the soft labels around p differ mostly because the step's INPUTS (value maps of earlier frames) already differed

Usage: python tools/davis_ref_agreement.py {18|50} [T]   ->  JSON on stdout (commit under profiles/)
"""
import json
import os
import runpy
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
import gen_golden as GG            # noqa: E402
import gen_davis_golden as GD      # noqa: E402
from oracle import exact_oracle as X          # noqa: E402
from oracle.vfs_oracle import fill_state_dict_  # noqa: E402

TOL_AFF = 5e-5      # cosine units: fp32 dot products of unit vectors (C <= 1024) + 1e-5 relative feature differences
QDIFF = 1e-4        # a query's soft label counts as different above this (a top-10 membership flip moves it by its softmax weight)


def footprint(val_q, h, w, H, W):
    """per output pixel: max (or any, for bool input) of val_q over the four feature positions its bilinear (align_corners=False) sample touches"""
    ys = (np.arange(H) + 0.5) * (h / H) - 0.5
    xs = (np.arange(W) + 0.5) * (w / W) - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, h - 1)
    y1 = np.clip(np.floor(ys).astype(int) + 1, 0, h - 1)
    x0 = np.clip(np.floor(xs).astype(int), 0, w - 1)
    x1 = np.clip(np.floor(xs).astype(int) + 1, 0, w - 1)
    m = val_q.reshape(h, w)
    return np.maximum(np.maximum(m[y0][:, x0], m[y0][:, x1]), np.maximum(m[y1][:, x0], m[y1][:, x1]))


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    builder, trackers, common = GG.import_reference_hot_path()
    gold = np.load(os.path.join(REPO, 'tests', 'golden', f'forward_test_r{depth}_davis.npz'))
    T = int(sys.argv[2]) if len(sys.argv) > 2 else gold['seg_preds'].shape[0]
    torch.set_num_threads(8)
    cfg = runpy.run_path(os.path.join(GG.REF, 'configs', GD.CFG[depth]))
    tc = GG.AttrDict(cfg['test_cfg'])
    bb = dict(cfg['model']['backbone'])
    bb['out_indices'], bb['strides'] = tc['out_indices'], tc['strides']
    model = builder.build_model(dict(type='VanillaTracker', backbone=bb), train_cfg=None, test_cfg=tc)
    fill_state_dict_(model, seed=5)
    model.eval()
    imgs, seg = GD.davis_clip(gold['seg_preds'].shape[0])
    imgs = imgs[:, :, :, :T]
    H, W = seg.shape
    radius, topk, temp, pre = int(tc['neighbor_range']) // 2, int(tc['topk']), float(tc['temperature']), int(tc['precede_frames'])

    # ---- the oracle's own free run (its features, its soft labels)
    t0 = time.time()
    sd = model.state_dict()
    frames = np.ascontiguousarray(np.transpose(imgs[0, 0].numpy(), (1, 0, 2, 3)))
    featB = np.concatenate([X.resnet_eval(sd, depth, frames[i:i + 10], strides=tuple(tc['strides']), out_indices=(2,), prefix='backbone.')[2]
                            for i in range(0, T, 10)])
    predsB, sbankB, bankB = X.propagate(featB, seg, (H, W), precede_frames=pre, topk=topk, temperature=temp, neighbor_range=tc['neighbor_range'],
                                        return_logits=True)
    _, h, w, C = featB.shape
    print(f'oracle run {time.time() - t0:.0f} s', file=sys.stderr)

    # ---- the reference, with its attention wrapped
    vt = sys.modules['mmaction.models.trackers.vanilla_tracker']
    orig = vt.masked_attention_efficient
    rows, state = [], dict(f=0, feat_rel=0.0)

    def wrapped(query, key, value, mask, **kw):
        outA = orig(query, key, value, mask, **kw)
        state['f'] += 1
        f = state['f']
        slots = [0] + list(range(max(0, f - pre), f))
        Tk = key.shape[2]
        assert Tk == len(slots)
        CO = value.shape[1]
        oA = outA[0].reshape(CO, h * w).t().contiguous().numpy()
        # feature agreement (reference vs oracle convolutions) on the query frame
        qA = query[0].permute(1, 2, 0).reshape(h * w, C).numpy()
        state['feat_rel'] = max(state['feat_rel'], float(np.abs(qA - featB[f].reshape(h * w, C)).max() / np.abs(qA).max()))
        # teacher-forced step: the oracle's features / arithmetic on the reference's value maps
        sb = np.zeros((T, h * w, CO), np.float32)
        vA = value[0].permute(1, 2, 3, 0).reshape(Tk, h * w, CO).numpy()
        for i, s in enumerate(slots):
            sb[s] = vA[i]
        oTF = X.labelprop(bankB, sb, f, slots, h, w, radius, topk, temp)
        qbad = np.abs(oA - oTF).max(1) > QDIFF
        # float64 near-tie check of those queries on the REFERENCE's features
        gaps = np.zeros(0)
        if qbad.any():
            kn = F.normalize(key[0].double(), p=2, dim=0).reshape(C, -1)
            qn = F.normalize(query[0].double(), p=2, dim=0).reshape(C, -1)[:, torch.from_numpy(np.nonzero(qbad)[0])]
            sc = kn.t() @ qn
            if mask is not None:
                sc.masked_fill_(~mask[:, torch.from_numpy(np.nonzero(qbad)[0])].bool().repeat(Tk, 1), float('-inf'))
            # the doubled first frame (vanilla_tracker.py:133-149) puts PAIRS of identical keys into the set: when such a pair sits on the
            # boundary (10th == 11th exactly) the membership question is whether its neighbours overtake the pair
            top = sc.topk(topk + 2, dim=0)[0]
            g0 = top[topk - 1] - top[topk]
            gaps = torch.where(g0 > 0, g0, torch.minimum(top[topk - 2] - top[topk - 1], top[topk] - top[topk + 1])).numpy()
        labA = X.seg_postprocess(oA, h, w, H, W)
        labTF = X.seg_postprocess(oTF, h, w, H, W)
        ref_map = gold['seg_preds'][f]
        post_diff = int((labA != ref_map).sum())       # the reference's own interpolate / min-max / argmax vs the oracle's on the SAME soft labels
        free = ref_map != predsB[f]
        tf_pix = labA != labTF
        fp = footprint(qbad, h, w, H, W)
        d_tf = footprint(np.abs(oA - oTF).max(1), h, w, H, W)
        d_in = footprint(np.abs(oA - sbankB[f]).max(1), h, w, H, W)
        # the reference's own float64 argmax margins at the mismatching pixels
        sel = free | tf_pix | (labA != ref_map)
        margin = np.zeros((H, W))
        r_min = 1.0
        if sel.any():
            up = F.interpolate(outA.double(), size=(H, W), mode='bilinear', align_corners=False)[0]
            mn, mx = up.reshape(CO, -1).min(1)[0].view(CO, 1, 1), up.reshape(CO, -1).max(1)[0].view(CO, 1, 1)
            r_min = float(torch.where(mx > 0, mx - mn, torch.ones_like(mx)).min())
            up = torch.where(mx > 0, (up - mn) / (mx - mn + 1e-12), up)
            idx = torch.from_numpy(np.nonzero(sel.reshape(-1))[0])
            t2 = up.reshape(CO, -1)[:, idx].topk(2, dim=0)[0]
            margin.reshape(-1)[idx.numpy()] = (t2[0] - t2[1]).numpy()
        # the min-max normalisation couples every pixel to the channel's extremes over the WHOLE map: a soft-label difference at the
        # pixel that holds a channel's maximum rescales that channel everywhere
        def extremes(o):
            u = F.interpolate(torch.from_numpy(np.ascontiguousarray(o.T)).double().reshape(1, CO, h, w), size=(H, W), mode='bilinear', align_corners=False)[0].reshape(CO, -1)
            return u.min(1)[0], u.max(1)[0]
        (mnA, mxA), (mnB, mxB), (mnT, mxT) = extremes(oA), extremes(sbankB[f]), extremes(oTF)
        g_in = float(((mnA - mnB).abs() + (mxA - mxB).abs()).max())
        g_tf = float(((mnA - mnT).abs() + (mxA - mxT).abs()).max())
        explained = margin <= 4.0 * (d_in + g_in) / r_min + 1e-6
        c_topk = free & explained & fp
        c_round = free & explained & ~fp & (d_tf >= 0.5 * d_in)
        c_prop = free & explained & ~fp & (d_tf < 0.5 * d_in)
        c_unexpl = free & ~explained
        tf_unexpl = tf_pix & ~(margin <= 4.0 * (d_tf + g_tf) / r_min + 1e-6)      # the same necessary condition for the teacher-forced flips
        rows.append(dict(frame=f, keys=Tk, free_mismatch=int(free.sum()), teacher_forced_mismatch=int(tf_pix.sum()),
                         queries_differ=int(qbad.sum()), max_gap_10_11=float(gaps.max()) if gaps.size else 0.0,
                         gaps_over_tol=int((gaps >= TOL_AFF).sum()), postprocess_mismatch=post_diff,
                         max_margin_of_a_flip=float(margin[free].max()) if free.any() else 0.0, r_min=r_min,
                         topk_tie=int(c_topk.sum()), rounding=int(c_round.sum()), propagated=int(c_prop.sum()), unexplained=int(c_unexpl.sum()),
                         teacher_forced_unexplained=int(tf_unexpl.sum())))
        print(rows[-1], file=sys.stderr)
        return outA

    vt.masked_attention_efficient = wrapped
    t0 = time.time()
    with torch.no_grad():
        res = model(imgs, return_loss=False, ref_seg_map=torch.from_numpy(seg)[None], img_meta=[dict(original_shape=(H, W, 3))])
    vt.masked_attention_efficient = orig
    assert np.array_equal(res[0].astype(np.uint8), gold['seg_preds'][:T]), 'the wrapped run must reproduce the committed golden maps'
    tot = {k: int(sum(r[k] for r in rows)) for k in ('free_mismatch', 'teacher_forced_mismatch', 'queries_differ', 'gaps_over_tol', 'postprocess_mismatch',
                                                       'topk_tie', 'rounding', 'propagated', 'unexplained', 'teacher_forced_unexplained')}
    npix = T * H * W
    out = dict(model=f'ResNet-{depth}', clip=[T, H, W], feature_map=[h, w, C], radius=radius, precede_frames=pre,
               agreement=1.0 - tot['free_mismatch'] / npix, pixels=npix, feature_max_rel_diff=state['feat_rel'],
               tolerances=dict(affinity_gap=TOL_AFF, query_differs=QDIFF),
               max_gap_10_11=max(r['max_gap_10_11'] for r in rows), max_margin_of_a_flip=max(r['max_margin_of_a_flip'] for r in rows),
               totals=tot, per_frame=rows, seconds=round(time.time() - t0))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
