"""Deterministic stand-in weights for benchmarks on a box without checkpoints (there is no network: the
reference's released checkpoints, README.md:73-74, cannot be fetched).  NOT an initialiser of the reference
(`init_weights` is that): the values only have to be non-degenerate - BatchNorm scales around one (the
zero-initialised last BatchNorm of every block included), non-trivial running statistics, convolution weights at
a variance-preserving scale - so that a timed evaluation pass propagates real labels instead of zeros.

Independent of the test oracle on purpose: `bench.py`'s GPU legs must not import `oracle/`."""
import math
import zlib

import torch


def _uniform(shape, seed, lo, hi):
    g = torch.Generator().manual_seed(seed & 0x7fffffff)
    return torch.rand(tuple(shape), generator=g) * (hi - lo) + lo


@torch.no_grad()
def synthetic_weights_(module, seed=0):
    """fill every floating-point entry of module.state_dict() in place; the per-tensor stream is seeded by a CRC of the
    entry's NAME, so the result does not depend on the order of the entries"""
    for name, t in module.state_dict().items():
        if not t.is_floating_point() or name == 'iteration':
            continue
        s = (seed * 7919 + zlib.crc32(name.encode())) & 0x7fffffff
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'running_var':
            v = _uniform(t.shape, s, 0.7, 1.3)
        elif leaf == 'running_mean':
            v = _uniform(t.shape, s, -0.2, 0.2)
        elif t.ndim == 1 and leaf == 'weight':       # BatchNorm gamma
            v = _uniform(t.shape, s, 0.7, 1.3)
        elif leaf == 'bias':
            v = _uniform(t.shape, s, -0.1, 0.1)
        else:                                        # conv / linear weight: uniform with variance 2 / fan_in (He)
            fan_in = t[0].numel() if t.ndim > 1 else t.numel()
            a = math.sqrt(6.0 / fan_in)
            v = _uniform(t.shape, s, -a, a)
        t.copy_(v.to(t.device, t.dtype))
    return module
