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
import re
from collections import OrderedDict

import torch

# One rewrite table, read in both directions.  Each row: (ConvModule-style pattern, torchvision-style template,
# torchvision-style pattern, ConvModule-style template); `n` is bn|gn, `t` the tensor name (weight, running_mean, ...).
_BLK = r'(?P<blk>layer\d+\.\d+)'
_T = r'(?P<t>[^.]+)$'
_RULES = (
    (r'^conv1\.conv\.' + _T, 'conv1.{t}',
     r'^conv1\.' + _T, 'conv1.conv.{t}'),
    (r'^conv1\.(?P<n>bn|gn)\.' + _T, '{n}1.{t}',
     r'^(?P<n>bn|gn)1\.' + _T, 'conv1.{n}.{t}'),
    (r'^' + _BLK + r'\.downsample\.conv\.' + _T, '{blk}.downsample.0.{t}',
     r'^' + _BLK + r'\.downsample\.0\.' + _T, '{blk}.downsample.conv.{t}'),
    (r'^' + _BLK + r'\.downsample\.(?P<n>bn|gn)\.' + _T, '{blk}.downsample.1.{t}',
     r'^' + _BLK + r'\.downsample\.1\.' + _T, '{blk}.downsample.bn.{t}'),
    (r'^' + _BLK + r'\.conv(?P<i>\d)\.conv\.' + _T, '{blk}.conv{i}.{t}',
     r'^' + _BLK + r'\.conv(?P<i>\d)\.' + _T, '{blk}.conv{i}.conv.{t}'),
    (r'^' + _BLK + r'\.conv(?P<i>\d)\.(?P<n>bn|gn)\.' + _T, '{blk}.{n}{i}.{t}',
     r'^' + _BLK + r'\.(?P<n>bn|gn)(?P<i>\d)\.' + _T, '{blk}.conv{i}.{n}.{t}'),
)
_TO_TV = [(re.compile(a), b) for a, b, _, _ in _RULES]
_FROM_TV = [(re.compile(c), d) for _, _, c, d in _RULES]


def _rewrite(table, key):
    for pat, template in table:
        m = pat.match(key)
        if m:
            return template.format(**m.groupdict())
    return None


def to_pretrained_keys(state_dict, verbose=False):
    """`backbone.*` entries renamed to torchvision keys; everything else is dropped; a backbone entry no row of the
    table covers raises RuntimeError (convert_to_pretrained.py raises on unknown sub-modules too)"""
    out = OrderedDict()
    for k, v in state_dict.items():
        if not k.startswith('backbone'):
            continue
        inner = k.replace('backbone.', '')
        name = _rewrite(_TO_TV, inner)
        if name is None:
            raise RuntimeError(inner)
        out[name] = v
        if verbose:
            print(f'{k} --> {name}')
    return out


def from_pretrained_keys(state_dict, prefix='backbone.'):
    """torchvision-style ResNet keys -> ConvModule-style keys (fc.* and unknown entries are skipped)"""
    out = OrderedDict()
    for k, v in state_dict.items():
        name = _rewrite(_FROM_TV, k)
        if name is not None:
            out[prefix + name] = v
    return out


def convert(src, dst):
    """tools/convert_weights/convert_to_pretrained.py: file -> file"""
    src_dict = torch.load(src, map_location='cpu')
    checkpoint = {'state_dict': to_pretrained_keys(src_dict.get('state_dict', src_dict), verbose=True), 'meta': {}}
    torch.save(checkpoint, dst)
    return checkpoint
