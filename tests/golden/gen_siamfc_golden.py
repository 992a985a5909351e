#!/usr/bin/env python3
"""Golden responses of the reference's SiamFC heads (projects/siamfc-pytorch/siamfc/heads.py, pure torch: loaded as
it is, in the build container only).  Inputs / weights from the closed-form fillers of oracle/vfs_oracle.py.
Usage: python tests/golden/gen_siamfc_golden.py  (writes tests/golden/siamfc_heads.npz)"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.vfs_oracle import fill_state_dict_, fill_tensor  # noqa: E402


def main():
    spec = importlib.util.spec_from_file_location('ref_heads', '/root/reference/projects/siamfc-pytorch/siamfc/heads.py')
    heads = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(heads)
    out = {}
    with torch.no_grad():
        # tracking shapes: one exemplar, three scales (siamfc_tracker_base.py:250-266), and a training-style batch
        for tag, (nz, nx, c, hz, h) in {'track': (1, 3, 64, 6, 13), 'batch': (4, 4, 64, 5, 9), 'groups': (2, 6, 128, 3, 8)}.items():
            z, x = fill_tensor([nz, c, hz, hz], 3, scale=1.5), fill_tensor([nx, c, h, h + 2], 4, scale=1.5)
            out[tag + '/siamfc'] = heads.SiamFC(out_scale=0.001)(z, x).numpy()
            head = heads.SiamConvFC(c, 2 * c, out_scale=0.01)
            fill_state_dict_(head, seed=21)
            out[tag + '/siamconvfc'] = head(z, x).numpy()
            out[tag + '/keys'] = np.array(list(head.state_dict().keys()))
            out[tag + '/shape'] = np.array([nz, nx, c, hz, h])
    np.savez_compressed(os.path.join(os.environ.get('VFS_GOLDEN_OUT', HERE), 'siamfc_heads.npz'), **out)
    print({k: v.shape for k, v in out.items() if 'siam' in k})


if __name__ == '__main__':
    main()
