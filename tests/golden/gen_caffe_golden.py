"""Golden stage outputs of the reference ResNet-50 with style='caffe' (the stride on the first 1x1 conv of a Bottleneck instead of
the 3x3, resnet.py:156-161), train mode, captured from the REAL reference class in the build container.

    python tests/golden/gen_caffe_golden.py        -> tests/golden/resnet50_caffe_fwd.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G      # noqa: E402
from gen_golden import fill_state_dict_, fill_tensor      # noqa: E402


def main():
    torch.manual_seed(0)
    G.import_reference_hot_path()
    from mmaction.models.backbones.resnet import ResNet
    net = ResNet(depth=50, pretrained=None, out_indices=(0, 1, 2, 3), style='caffe',
                 norm_cfg=dict(type='SyncBN', requires_grad=True), zero_init_residual=True)
    net.init_weights()
    fill_state_dict_(net, seed=50)
    net.train()
    outs = net(fill_tensor([2, 3, 64, 64], seed=7, scale=2.0))
    assert net.layer2[0].conv1.conv.stride == (2, 2) and net.layer2[0].conv2.conv.stride == (1, 1)
    np.savez_compressed(os.path.join(os.environ.get('VFS_GOLDEN_OUT', HERE), 'resnet50_caffe_fwd.npz'),
                        **{f'sample{i}': o.detach().flatten()[::7].numpy().copy() for i, o in enumerate(outs)},
                        **{f'shape{i}': np.array(o.shape) for i, o in enumerate(outs)},
                        keys=np.array(list(net.state_dict().keys())))
    print('wrote resnet50_caffe_fwd', [tuple(o.shape) for o in outs])


if __name__ == '__main__':
    main()
