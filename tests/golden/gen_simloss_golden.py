"""Golden vectors for CosineSimLoss outside the shipped configs' [N,C] case - pairwise affinity (with / without mask),
with_norm=False, spatial non-pairwise operands - captured from the REAL reference class
(mmaction/models/losses/sim_loss.py:25-63) in the build container.

    python tests/golden/gen_simloss_golden.py        -> tests/golden/simloss_pairwise.npz

Per case: loss [B] and the gradients of sum_b w_b * loss_b wrt both operands (w = fill_tensor([B], 9) + 1.5)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G      # noqa: E402
from gen_golden import fill_tensor      # noqa: E402

CASES = [   # name, shape of cls_score, shape of label (None = same), kwargs, mask?
    ('pair_mask', [3, 48, 5, 7], None, dict(pairwise=True), True),
    ('pair_nomask_neg', [2, 70, 6, 6], None, dict(pairwise=True, negative=True), False),
    ('pair_nonorm_mask', [2, 33, 40], None, dict(pairwise=True, with_norm=False, loss_weight=0.5), True),
    ('pair_rect', [2, 64, 37], [2, 64, 50], dict(pairwise=True), True),
    ('spatial_nonpair', [3, 40, 4, 9], None, dict(), False),
    ('spatial_nonpair_nonorm_neg', [2, 20, 35], None, dict(with_norm=False, negative=True), False),
    ('rows_nonorm', [6, 32], None, dict(with_norm=False), False),
]


def main():
    G.import_reference_hot_path()
    from mmaction.models.losses.sim_loss import CosineSimLoss
    res = {}
    for name, sa, sl, kw, use_mask in CASES:
        sl = sl or sa
        a = fill_tensor(sa, 101, scale=1.5).requires_grad_(True)
        l = fill_tensor(sl, 102, scale=1.5).requires_grad_(True)
        mask = None
        if use_mask:
            Sa, Sl = int(np.prod(sa[2:])), int(np.prod(sl[2:]))
            mask = (fill_tensor([sa[0], Sa, Sl], 103) > -0.2)
        loss = CosineSimLoss(**kw)(a, l, mask=mask) if mask is not None else CosineSimLoss(**kw)(a, l)
        w = fill_tensor([sa[0]], 9) + 1.5
        (loss * w).sum().backward()
        res[name + '/loss'] = loss.detach().numpy()
        res[name + '/da'] = a.grad.numpy()
        res[name + '/dl'] = l.grad.numpy()
        print(name, loss.detach().numpy())
    np.savez_compressed(os.path.join(os.environ.get('VFS_GOLDEN_OUT', HERE), 'simloss_pairwise.npz'), **res)


if __name__ == '__main__':
    main()
