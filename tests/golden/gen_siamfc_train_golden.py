#!/usr/bin/env python3
"""Golden vectors of the SiamFC probe's TRAINING step, produced by the reference's own code in the build container:
`heads.SiamConvFC`, `losses.BalancedLoss` / `losses.FocalLoss` (projects/siamfc-pytorch/siamfc, pure torch: loaded as
they are) and `TrackerSiamFC._create_labels` (siamfc_tracker_base.py:456-500; the module imports cv2 / got10k, which are
absent, so the method's own source is compiled out of the file and run with a stand-in `self`), torch.optim.Adam / SGD
as siamfc_tracker_base.py:131-147 builds them.  Inputs / weights from the closed-form fillers of oracle/vfs_oracle.py.
Usage: python tests/golden/gen_siamfc_train_golden.py  (writes tests/golden/siamfc_train.npz)"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.vfs_oracle import fill_state_dict_, fill_tensor  # noqa: E402

REF = '/root/reference/projects/siamfc-pytorch/siamfc'


def load(name):
    spec = importlib.util.spec_from_file_location('ref_' + name, os.path.join(REF, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_create_labels():
    """the reference's TrackerSiamFC._create_labels, compiled from its own source text (not copied anywhere)"""
    tree = ast.parse(open(os.path.join(REF, 'siamfc_tracker_base.py')).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'TrackerSiamFC'][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == '_create_labels'][0]
    ns = {'np': np, 'torch': torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'siamfc_tracker_base.py', 'exec'), ns)
    return ns['_create_labels']


def main():
    heads, losses = load('heads'), load('losses')
    create_labels = reference_create_labels()
    out = {}
    nz, c, hz, h = 4, 64, 5, 12
    zf, xf = fill_tensor([nz, c, hz, hz], 3, scale=1.5), fill_tensor([nz, c, h, h], 4, scale=1.5)
    for tag, crit, optname in (('focal_adam', losses.FocalLoss(), 'Adam'), ('balance_sgd', losses.BalancedLoss(), 'SGD'),
                               ('balance_adam_wd', losses.BalancedLoss(), 'AdamWD')):
        head = heads.SiamConvFC(c, c, out_scale=0.01)
        fill_state_dict_(head, seed=21)
        params = list(head.parameters())
        if optname == 'SGD':
            opt = torch.optim.SGD(params, lr=1e-2, weight_decay=5e-4, momentum=0.9)
        else:
            opt = torch.optim.Adam(params, lr=1e-3, weight_decay=5e-4 if optname == 'AdamWD' else 0)
        me = types.SimpleNamespace(cfg=types.SimpleNamespace(r_pos=16, r_neg=0, total_stride=8), device=torch.device('cpu'))
        for step in range(2):
            resp = head(zf, xf)
            labels = create_labels(me, resp.size())
            loss = crit(resp, labels)
            opt.zero_grad()
            loss.backward()
            if step == 0:
                out[tag + '/responses'] = resp.detach().numpy().copy()
                out[tag + '/labels'] = labels.numpy().copy()
                out[tag + '/loss'] = np.float64(loss.item())
                for n, p in head.named_parameters():
                    out[f'{tag}/grad/{n}'] = p.grad.numpy().copy()
            else:
                out[tag + '/loss2'] = np.float64(loss.item())
            opt.step()
            for n, p in head.named_parameters():
                out[f'{tag}/step{step + 1}/{n}'] = p.detach().numpy().copy()
    # labels with a soft ring (r_neg > r_pos): exercises the 0.5 entries BalancedLoss ignores
    me = types.SimpleNamespace(cfg=types.SimpleNamespace(r_pos=16, r_neg=32, total_stride=8), device=torch.device('cpu'))
    lab = create_labels(me, torch.Size([2, 1, 9, 9]))
    x = fill_tensor([2, 1, 9, 9], 7, scale=3.0).requires_grad_(True)
    for tag, crit in (('ring/balance', losses.BalancedLoss(neg_weight=0.5)), ('ring/focal', losses.FocalLoss(gamma=1.5))):
        x.grad = None
        loss = crit(x, lab)
        loss.backward()
        out[tag + '/loss'] = np.float64(loss.item())
        out[tag + '/grad'] = x.grad.numpy().copy()
    out['ring/labels'] = lab.numpy().copy()
    np.savez_compressed(os.path.join(os.environ.get('VFS_GOLDEN_OUT', HERE), 'siamfc_train.npz'), **out)
    print({k: getattr(v, 'shape', v) for k, v in out.items() if 'grad' not in k and 'step' not in k})


if __name__ == '__main__':
    main()
