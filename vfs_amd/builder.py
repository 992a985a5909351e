"""Factory entry points under the names the reference's callers use
(mmaction/models/builder.py:8-78: tools/train.py:152 and tools/test.py:133 call build_model;
trackers and heads call build_backbone / build_head / build_loss on their nested dicts)."""
import torch.nn as nn

from . import registry as R


def build(cfg, registry, default_args=None):
    """A dict builds one module; a list of dicts builds an nn.Sequential of them."""
    if isinstance(cfg, list):
        return nn.Sequential(*(R.build_from_cfg(c, registry, default_args) for c in cfg))
    return R.build_from_cfg(cfg, registry, default_args)


def _plain(registry):
    return lambda cfg: build(cfg, registry)


def _with_cfgs(registry):
    def fn(cfg, train_cfg=None, test_cfg=None):
        return build(cfg, registry, dict(train_cfg=train_cfg, test_cfg=test_cfg))
    return fn


build_backbone = _plain(R.BACKBONES)
build_head = _plain(R.HEADS)
build_loss = _plain(R.LOSSES)
build_drop_layer = _plain(R.DROP_LAYERS)
build_localizer = _plain(R.LOCALIZERS)
build_recognizer = _with_cfgs(R.RECOGNIZERS)
build_tracker = _with_cfgs(R.TRACKERS)


def build_model(cfg, train_cfg=None, test_cfg=None):
    """Dispatch on which registry owns cfg['type'] (localizers take no train/test cfg)."""
    kind = cfg['type']
    for reg, fn in ((R.LOCALIZERS, lambda: build_localizer(cfg)),
                    (R.RECOGNIZERS, lambda: build_recognizer(cfg, train_cfg, test_cfg)),
                    (R.TRACKERS, lambda: build_tracker(cfg, train_cfg, test_cfg))):
        if kind in reg:
            return fn()
    raise KeyError(f'{kind} not in any registry')
