"""Checkpoint key conventions of the reference.

* `to_pretrained_keys` / `convert`: a VFS training checkpoint (`backbone.layerX.Y.convN.{conv,bn}.*`,
  mmcv ConvModule naming) -> torchvision-style ResNet keys (`layerX.Y.convN.*`, `layerX.Y.bnN.*`,
  `downsample.{0,1}.*`), the behaviour of tools/convert_weights/convert_to_pretrained.py:7-64
  (non-backbone entries are dropped, unknown sub-modules raise RuntimeError, the result is saved as
  {'state_dict': ..., 'meta': {}}).
* `from_pretrained_keys`: the inverse mapping, i.e. what ResNet._load_torchvision_checkpoint
  (models/backbones/resnet.py:488-523) does while loading; `vfs_amd.resnet.ResNet.
  load_torchvision_checkpoint` applies it module by module.
"""
from collections import OrderedDict

import torch

_NORMS = {'bn': 'bn', 'gn': 'gn'}


def to_pretrained_keys(state_dict, verbose=False):
    out = OrderedDict()
    for k, v in state_dict.items():
        if not k.startswith('backbone'):
            continue
        b_k = k.replace('backbone.', '')
        parts = b_k.split('.')
        tail = parts[-1]
        if b_k.startswith('conv1'):
            if parts[1] == 'conv':
                name = f'conv1.{tail}'
            elif parts[1] in _NORMS:
                name = f'{_NORMS[parts[1]]}1.{tail}'
            else:
                raise RuntimeError(b_k)
        elif b_k.startswith('layer'):
            layer, block = int(parts[0][-1]), int(parts[1])
            if parts[2] == 'downsample':
                if parts[3] == 'conv':
                    name = f'layer{layer}.{block}.downsample.0.{tail}'
                elif parts[3] in _NORMS:
                    name = f'layer{layer}.{block}.downsample.1.{tail}'
                else:
                    raise RuntimeError(b_k)
            elif parts[3] == 'conv':
                name = f'layer{layer}.{block}.conv{int(parts[2][-1])}.{tail}'
            elif parts[3] in _NORMS:
                name = f'layer{layer}.{block}.{_NORMS[parts[3]]}{int(parts[2][-1])}.{tail}'
            else:
                raise RuntimeError(b_k)
        else:
            raise RuntimeError(f'{b_k}')
        out[name] = v
        if verbose:
            print(f'{k} --> {name}')
    return out


def from_pretrained_keys(state_dict, prefix='backbone.'):
    """torchvision-style ResNet keys -> ConvModule-style keys (fc.* and unknown entries are skipped)"""
    out = OrderedDict()
    for k, v in state_dict.items():
        parts = k.split('.')
        tail = parts[-1]
        if parts[0] == 'conv1':
            name = f'conv1.conv.{tail}'
        elif parts[0] in ('bn1', 'gn1'):
            name = f'conv1.{parts[0][:2]}.{tail}'
        elif parts[0].startswith('layer') and len(parts) >= 4:
            head = f'{parts[0]}.{parts[1]}'
            if parts[2] == 'downsample':
                name = f'{head}.downsample.{"conv" if parts[3] == "0" else "bn"}.{tail}'
            elif parts[2].startswith('conv'):
                name = f'{head}.{parts[2]}.conv.{tail}'
            elif parts[2][:2] in ('bn', 'gn'):
                name = f'{head}.conv{parts[2][2:]}.{parts[2][:2]}.{tail}'
            else:
                continue
        else:
            continue
        out[prefix + name] = v
    return out


def convert(src, dst):
    """tools/convert_weights/convert_to_pretrained.py: file -> file"""
    src_dict = torch.load(src, map_location='cpu')
    checkpoint = {'state_dict': to_pretrained_keys(src_dict.get('state_dict', src_dict), verbose=True), 'meta': {}}
    torch.save(checkpoint, dst)
    return checkpoint
