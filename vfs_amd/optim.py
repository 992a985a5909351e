"""Fused SGD over the tracker's flat parameter arena (torch.optim.SGD semantics of configs/r*_*.py:134: lr, momentum,
weight_decay; dampening 0): one HIP launch per contiguous range of TRAINABLE parameters (one launch for the shipped
configs).  Parameters with requires_grad=False are never touched - the reference's optimizer does not hold them, so
they get neither weight decay nor momentum.  The momentum arena is exposed through `state[p]['momentum_buffer']`
(views), so `state_dict()` / `load_state_dict()` - what mmcv's checkpoint hook and `--resume-from` use - carry it."""
import os

import torch

from .engine import bump_params_epoch, shared_engine


class SGD(torch.optim.Optimizer):
    def __init__(self, model, lr=0.05, momentum=0.9, weight_decay=1e-4):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._buf = None
        self._segments = None

    def zero_grad(self, set_to_none=False):
        f = self.model._ensure_arena()
        f['grads'].zero_()

    def _arena(self):
        """momentum arena + the contiguous arena ranges that hold trainable parameters"""
        f = self.model._ensure_arena()
        flat = f['params']
        if self._buf is None or self._buf.shape != flat.shape or self._buf.device != flat.device:
            old = {id(p): st.get('momentum_buffer') for p, st in self.state.items()}
            self._buf = torch.zeros_like(flat)
            self._segments = None
            for p, o in zip(f['plist'], f['offsets']):
                if not p.requires_grad:
                    continue
                view = self._buf[o:o + p.numel()].view(p.shape)
                prev = old.get(id(p))
                if prev is not None:              # a state restored before the arena existed
                    view.copy_(prev)
                self.state[p]['momentum_buffer'] = view
        key = tuple(p.requires_grad for p in f['plist'])
        if self._segments is None or self._segments[0] != key:
            segs = []
            for p, o in zip(f['plist'], f['offsets']):
                if not p.requires_grad:
                    continue
                end = o + (p.numel() + 3) // 4 * 4
                if segs and segs[-1][1] == o:
                    segs[-1][1] = end
                else:
                    segs.append([o, end])
            self._segments = (key, segs)
        return f, self._segments[1]

    @torch.no_grad()
    def step(self, closure=None):
        f, segs = self._arena()
        flat, g = f['params'], f['grads']
        grp = self.param_groups[0]
        eng = shared_engine()
        bump_params_epoch()      # raw-pointer update: caches derived from the parameters (vfs_amd/exact.py) must refresh
        # data parallel with the SyncBN window exchange: its error word gates the update on the device (a peer that never arrived
        # poisons the step's statistics with NaN; the host only learns of it when it reads the log values)
        x = eng._p2p
        skip = x.state[1:2] if x is not None and x.state.device == flat.device else None
        for lo, hi in segs:
            eng.timed('sgd', (0.0, 20.0 * (hi - lo)), flat.device, eng.lib.sgd_step, flat[lo:hi], g[lo:hi], self._buf[lo:hi], hi - lo,
                      float(grp['lr']), float(grp['momentum']), float(grp['weight_decay']), skip, eng.stream(flat.device))
        if skip is not None:
            self._watch_exchange(x)

    def _watch_exchange(self, x, every=int(os.environ.get('VFS_P2P_CHECK_EVERY', '50'))):
        """the error word of the SyncBN window exchange is sticky: once set, every later update is skipped on every rank (the word is
        MAX-reduced over the ranks, trackers.py).  A consumer that never reads the log values would then train as a silent no-op
        (advisor r05): every `every` steps a copy of the word is queued to pinned memory and the copy queued `every` steps earlier -
        long complete, no stall - is looked at; a set word raises here, at most 2 x `every` steps after the failed exchange."""
        self._xsteps = getattr(self, '_xsteps', 0) + 1
        if self._xsteps % every:
            return
        pend = getattr(self, '_xpending', None)
        if pend is not None:
            pend[1].synchronize()
            if int(pend[0][0]):
                raise RuntimeError('SyncBN P2P exchange: a peer did not arrive within the spin limit (VFS_P2P_SPIN) in an earlier step; every '
                                   'update since has been skipped on all ranks - stop, or restart from the last checkpoint')
        host = torch.empty(1, dtype=torch.int64, pin_memory=True)
        host.copy_(x.state[1:2], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._xpending = (host, ev)

    def load_state_dict(self, state_dict):
        """torch's loader replaces the state tensors by copies: put them back into the momentum arena"""
        super().load_state_dict(state_dict)
        loaded = {id(p): st.get('momentum_buffer') for p, st in self.state.items()}
        self._buf = None
        f, _ = self._arena()
        for p in f['plist']:
            buf = loaded.get(id(p))
            if buf is not None and p.requires_grad:
                self.state[p]['momentum_buffer'].copy_(buf)


def build_optimizer(model, cfg):
    """mmcv build_optimizer for the shipped `optimizer = dict(type='SGD', ...)`."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    if t != 'SGD':
        raise KeyError(f'optimizer type {t} is not on the VFS path')
    return SGD(model, **cfg)
