"""DAVIS-2017 semi-supervised evaluation on the GPU: counterpart of DavisDataset.davis_evaluate /
.evaluate (reference mmaction/datasets/davis_dataset.py:68-181).  The reference writes palette PNGs
and calls the un-vendored `davis2017.evaluation.DAVISEvaluation`; here the label maps stay on the
device, `vfs_davis_counts` (csrc/davis.hip) produces the integer ingredients of J and F for every
object and frame, and the (tiny) per-object statistics are finished on the host in float64.

Returned keys are the reference's: 'J&F-Mean', 'J-Mean', 'J-Recall', 'J-Decay', 'F-Mean',
'F-Recall', 'F-Decay' (davis_dataset.py:109-140), prefixed 'feat_{i}.' for multi-feature results
(:159-176)."""
import math
import os

import numpy as np
import torch

from ._lib import get_lib

# davis_dataset.py:21-26
PALETTE = [[0, 0, 0], [128, 0, 0], [0, 128, 0], [128, 128, 0], [0, 0, 128], [128, 0, 128], [0, 128, 128],
           [128, 128, 128], [64, 0, 0], [191, 0, 0], [64, 128, 0], [191, 128, 0], [64, 0, 128], [191, 0, 128],
           [64, 128, 128], [191, 128, 128], [0, 64, 0], [128, 64, 0], [0, 191, 0], [128, 191, 0], [0, 64, 128],
           [128, 64, 128]]
BOUND_TH = 0.008
G_MEASURES = ['J&F-Mean', 'J-Mean', 'J-Recall', 'J-Decay', 'F-Mean', 'F-Recall', 'F-Decay']


def bound_pixels(h, w, bound_th=BOUND_TH):
    return int(bound_th) if bound_th >= 1 else int(math.ceil(bound_th * math.hypot(h, w)))


def _statistics(values):
    """mean, recall (> 0.5), decay (first minus last quarter) with the package's bin edges"""
    values = np.asarray(values, dtype=np.float64)
    n_bins = 4
    ids = (np.round(np.linspace(1, len(values), n_bins + 1) + 1e-10) - 1).astype(np.uint8)
    bins = [values[ids[i]:ids[i + 1] + 1] for i in range(n_bins)]
    return float(np.nanmean(values)), float(np.nanmean(values > 0.5)), float(np.nanmean(bins[0]) - np.nanmean(bins[3]))


def _jf_from_counts(c):
    """c: int64 [F][6] of one object -> (J per frame, F per frame)"""
    c = c.astype(np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        j = np.where(c[:, 1] == 0, 1.0, c[:, 0] / c[:, 1])
        prec = np.where(c[:, 2] == 0, 1.0, c[:, 4] / c[:, 2])
        rec = np.where(c[:, 3] == 0, 1.0, c[:, 5] / c[:, 3])
    prec = np.where((c[:, 2] > 0) & (c[:, 3] == 0), 0.0, prec)
    rec = np.where((c[:, 2] == 0) & (c[:, 3] > 0), 0.0, rec)
    with np.errstate(divide='ignore', invalid='ignore'):
        f = np.where(prec + rec == 0, 0.0, 2 * prec * rec / (prec + rec))
    return j, f


def sequence_counts(pred, gt, device=None, bound_th=BOUND_TH, use_void=False, nobj=None):
    """pred, gt: uint8 [T,H,W] (tensor on the GPU, or array-like that is uploaded) ->
    int64 numpy [K][T-2][6]."""
    lib = get_lib()
    if not torch.is_tensor(gt):
        gt = torch.as_tensor(np.ascontiguousarray(gt))
    if not torch.is_tensor(pred):
        pred = torch.as_tensor(np.ascontiguousarray(pred))
    if device is None:
        device = pred.device if pred.is_cuda else (gt.device if gt.is_cuda else torch.device('cuda'))
    pred = pred.to(device=device, dtype=torch.uint8).contiguous()
    gt = gt.to(device=device, dtype=torch.uint8).contiguous()
    assert pred.shape == gt.shape and pred.dim() == 3, (pred.shape, gt.shape)
    T, H, W = gt.shape
    if nobj is None:
        ids = torch.unique(gt)
        ids = ids[(ids != 0) & (ids != 255)]
        nobj = int(ids.max()) if ids.numel() else 0
    F = max(T - 2, 0)
    if nobj == 0 or F == 0:
        return np.zeros((nobj, F, 6), dtype=np.int64)
    counts = torch.empty(F, nobj, 6, dtype=torch.int32, device=device)
    scratch = torch.empty(2 * F * H * W, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else None
    lib.davis_counts(pred, gt, counts, scratch, T, H, W, nobj, bound_pixels(H, W, bound_th), 1 if use_void else 0, stream)
    return counts.cpu().numpy().astype(np.int64).transpose(1, 0, 2).copy()


def evaluate_sequences(sequences, device=None, bound_th=BOUND_TH):
    """sequences: {name: (pred [T,H,W], gt [T,H,W])} -> metric dict + 'per_object' {name_k: (J, F)}."""
    jm, jr, jd, fm, fr, fd, per_obj = [], [], [], [], [], [], {}
    for name, (pred, gt) in sequences.items():
        counts = sequence_counts(pred, gt, device, bound_th)
        for k in range(counts.shape[0]):
            j, f = _jf_from_counts(counts[k])
            a, b, c = _statistics(j)
            d, e, g = _statistics(f)
            jm.append(a); jr.append(b); jd.append(c); fm.append(d); fr.append(e); fd.append(g)
            per_obj[f'{name}_{k + 1}'] = (a, d)
    mean = lambda v: float(np.mean(v)) if v else float('nan')   # noqa: E731
    res = {'J&F-Mean': (mean(jm) + mean(fm)) / 2.0, 'J-Mean': mean(jm), 'J-Recall': mean(jr), 'J-Decay': mean(jd),
           'F-Mean': mean(fm), 'F-Recall': mean(fr), 'F-Decay': mean(fd)}
    res['per_object'] = per_obj
    return res


class DavisEvaluator:
    """`evaluate(results, metrics='davis')` of the reference's dataset class, fed with ground-truth label
    maps instead of a davis_root directory: gts = list of uint8 [T,H,W] (one per video), names optional."""

    def __init__(self, gts, names=None, device=None):
        self.gts = list(gts)
        self.names = list(names) if names is not None else [f'video{i:03d}' for i in range(len(self.gts))]
        self.device = device

    def __len__(self):
        return len(self.gts)

    def davis_evaluate(self, results, output_dir=None, logger=None):
        assert len(results) == len(self)
        if output_dir is not None:
            save_palette_pngs(results, output_dir, self.names)
        seqs = {}
        for name, res, gt in zip(self.names, results, self.gts):
            assert len(res) == len(gt), (name, len(res), len(gt))
            seqs[name] = (res, gt)
        full = evaluate_sequences(seqs, self.device)
        self.per_object = full.pop('per_object')
        if logger is not None:
            logger.info('Global results: ' + ', '.join(f'{k} {full[k]:.4f}' for k in G_MEASURES))
        return full

    def evaluate(self, results, metrics='davis', output_dir=None, logger=None):
        metrics = metrics if isinstance(metrics, (list, tuple)) else [metrics]
        for metric in metrics:
            if metric not in ('davis',):
                raise KeyError(f'metric {metric} is not supported')
        out = {}
        first = results[0]
        if (isinstance(first, np.ndarray) and first.ndim == 4) or isinstance(first, list):   # several feature levels
            for fi in range(len(first)):
                part = self.davis_evaluate([r[fi] for r in results], output_dir, logger)
                out.update({f'feat_{fi}.{k}': v for k, v in part.items()})
        else:
            out.update(self.davis_evaluate(results, output_dir, logger))
        return out


def save_palette_pngs(results, output_dir, names, filename_tmpl='{:05}.png'):
    """davis_dataset.py:94-106: one 8-bit palette PNG per frame under output_dir/<video>/"""
    from PIL import Image
    pal = np.asarray(PALETTE, dtype=np.uint8).ravel()
    for name, res in zip(names, results):
        res = res.cpu().numpy() if torch.is_tensor(res) else np.asarray(res)
        d = os.path.join(output_dir, name)
        os.makedirs(d, exist_ok=True)
        for i in range(res.shape[0]):
            img = Image.fromarray(res[i].astype(np.uint8))
            img.putpalette(pal)
            img.save(os.path.join(d, filename_tmpl.format(i)))


def load_palette_pngs(video_dir, filename_tmpl='{:05}.png', num_frames=None):
    """inverse of save_palette_pngs for one video: uint8 [T,H,W] of palette indices"""
    from PIL import Image
    frames = []
    i = 0
    while num_frames is None or i < num_frames:
        path = os.path.join(video_dir, filename_tmpl.format(i))
        if not os.path.exists(path):
            break
        frames.append(np.array(Image.open(path)))
        i += 1
    return np.stack(frames).astype(np.uint8)
