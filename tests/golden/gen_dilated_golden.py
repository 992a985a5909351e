#!/usr/bin/env python3
"""Golden features of the reference ResNet built as the SiamFC probe builds it
(projects/siamfc-pytorch/siamfc/default_config_base.py:40-49: dilations (1,1,2,4), strides (1,2,1,1), frozen,
norm_eval), eval mode, run by the REAL reference class in the build container (mmcv stand-in of gen_golden.py).
Usage: python tests/golden/gen_dilated_golden.py  (writes tests/golden/resnet{18,50}_dilated_eval.npz)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference_hot_path  # noqa: E402
from oracle.vfs_oracle import fill_state_dict_, fill_tensor  # noqa: E402


def main():
    import_reference_hot_path()
    from mmaction.models.backbones.resnet import ResNet
    out_dir = os.environ.get('VFS_GOLDEN_OUT', HERE)
    for depth in (18, 50):
        net = ResNet(depth=depth, pretrained=None, out_indices=(3,), strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4),
                     frozen_stages=4, norm_eval=True, norm_cfg=dict(type='BN', requires_grad=True), zero_init_residual=False)
        net.init_weights()
        fill_state_dict_(net, seed=depth + 100)
        net.eval()
        x = fill_tensor([2, 3, 64, 80], seed=9, scale=2.0)
        with torch.no_grad():
            y = net(x)
        flat = y.flatten()
        np.savez_compressed(os.path.join(out_dir, f'resnet{depth}_dilated_eval.npz'), shape=np.array(y.shape),
                            sample=flat[::13].numpy().copy(),       # every 13th value + two checksums keep the fixture small
                            checksum=np.array([flat.double().sum().item(), flat.double().abs().sum().item()]),
                            keys=np.array(list(net.state_dict().keys())))
        print(depth, tuple(y.shape), float(y.abs().mean()))


if __name__ == '__main__':
    main()
