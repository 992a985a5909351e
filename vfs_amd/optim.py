"""Fused SGD over the tracker's flat parameter arena: one HIP launch per step
(torch.optim.SGD semantics of configs/r*_*.py:134: lr, momentum, weight_decay; dampening 0)."""
import torch

from .engine import shared_engine


class SGD(torch.optim.Optimizer):
    def __init__(self, model, lr=0.05, momentum=0.9, weight_decay=1e-4):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._buf = None

    def zero_grad(self, set_to_none=False):
        f = self.model._ensure_arena()
        f['grads'].zero_()

    @torch.no_grad()
    def step(self, closure=None):
        f = self.model._ensure_arena()
        flat, g = f['params'], f['grads']
        if self._buf is None or self._buf.shape != flat.shape or self._buf.device != flat.device:
            self._buf = torch.zeros_like(flat)
        grp = self.param_groups[0]
        eng = shared_engine()
        eng.lib.sgd_step(flat, g, self._buf, flat.numel(), float(grp['lr']), float(grp['momentum']),
                         float(grp['weight_decay']), eng.stream(flat.device))


def build_optimizer(model, cfg):
    """mmcv build_optimizer for the shipped `optimizer = dict(type='SGD', ...)`."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    if t != 'SGD':
        raise KeyError(f'optimizer type {t} is not on the VFS path')
    return SGD(model, **cfg)
