"""Trackers under the reference's registry names, constructors and call signatures
(mmaction/models/trackers/base.py, sim_siam_base_tracker.py, vanilla_tracker.py).

`forward_train` runs the whole two-view SimSiam step (frames -> NHWC4, ResNet, head, cosine
loss) as one chain of HIP kernels and returns the reference's loss dict of UNREDUCED [N]
tensors; `loss.backward()` (what mmcv's OptimizerHook calls) triggers the hand-written backward
chain through a single autograd node, which accumulates into `param.grad` and -- when
torch.distributed is initialised -- all-reduces the flat gradient arena over RCCL in buckets
that overlap the remaining backward kernels (the DDP reducer's job in the reference,
apis/train.py:62-66)."""
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import builder
from .engine import BF16, bump_params_epoch, shared_engine
from .registry import TRACKERS
from .resnet import ResNet


def add_prefix(inputs, prefix):
    """mmaction/utils/misc.py:30-46."""
    return {f'{prefix}.{k}': v for k, v in inputs.items()}


class _hp_chain:
    """context: run a launch chain on a high-priority stream, ordered after the caller's stream and joined back into
    it.  Used when the step contains collectives (N > 1; VFS_MAIN_PRIO=0/1 overrides): the many short kernels around
    the SyncBN all-reduces then win the dispatch against the weight-gradient stream - measured in a 1-rank RCCL
    group: R50 13.47 -> 12.55 ms, R18 9.40 -> 9.13; without collectives it changes nothing."""
    _streams = {}

    def __init__(self, dev):
        self.dev = dev
        want = os.environ.get('VFS_MAIN_PRIO')
        self.on = dev.type == 'cuda' and (want == '1' or (want is None and shared_engine().collectives_on))

    def __enter__(self):
        if not self.on:
            return self
        hp = self._streams.get(self.dev)
        if hp is None:
            hp = self._streams[self.dev] = torch.cuda.Stream(self.dev, priority=-1)
        self.cur = torch.cuda.current_stream(self.dev)
        hp.wait_stream(self.cur)
        self.ctx = torch.cuda.stream(hp)
        self.ctx.__enter__()
        self.hp = hp
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
            self.cur.wait_stream(self.hp)
        return False


class _TrainStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, tracker, imgs):
        ctx.tracker = tracker
        with _hp_chain(imgs.device):
            return tracker._step_forward(imgs)

    @staticmethod
    def backward(ctx, g):
        with _hp_chain(g.device):
            ctx.tracker._step_backward(g)
        return None, None, None


class _ReduceLossFn(torch.autograd.Function):
    """loss rows [K, N] -> the scalar 'loss' of _parse_losses, already reduced on the device by
    vfs_loss_means inside the forward chain; backward hands every row element grad / N."""

    @staticmethod
    def forward(ctx, rows, means):
        ctx.shape = rows.shape
        return means[-1].detach().clone()

    @staticmethod
    def backward(ctx, g):
        K, N = ctx.shape
        return (g * (1.0 / N)).expand(K, N), None


class LazyLogVars(OrderedDict):
    """log_vars of the fused train_step: same keys, same python floats as _parse_losses returns (base.py:103-108), but read
    from the device on FIRST ACCESS instead of inside train_step.  The reference's per-variable .item() stalls the host in
    the middle of the step; here the host goes on to enqueue zero_grad / backward / the optimizer while the forward still
    runs (0.2 ms of GPU idle time per ResNet-50 step otherwise), and whoever consumes the values - mmcv's log buffer, a
    print - waits, when it actually looks, only for the small copy that was queued behind the forward chain."""

    def __init__(self, keys, snapshot):
        super().__init__((k, None) for k in keys)
        self._snap = snapshot

    def _load(self):
        snap = self.__dict__.pop('_snap', None)
        if snap is not None:
            if isinstance(snap, tuple):       # (pinned host copy, event recorded behind the asynchronous copy[, P2P error word])
                snap[1].synchronize()
                if len(snap) > 2 and int(snap[2][0]):
                    raise RuntimeError('SyncBN P2P exchange: a peer did not arrive within the spin limit (VFS_P2P_SPIN); the '
                                       'BatchNorm statistics of this step are invalid')
                snap = snap[0]
            for k, v in zip(list(super().keys()), snap.tolist()):
                super().__setitem__(k, v)

    def __getitem__(self, k):
        self._load()
        return super().__getitem__(k)

    def get(self, k, default=None):
        self._load()
        return super().get(k, default)

    def items(self):
        self._load()
        return super().items()

    def values(self):
        self._load()
        return super().values()

    def __repr__(self):
        self._load()
        return super().__repr__()

    def __eq__(self, other):
        self._load()
        return dict(self) == dict(other)

    def __reduce__(self):
        self._load()
        return (OrderedDict, (list(self.items()),))

    def copy(self):
        self._load()
        return OrderedDict(self)

    def pop(self, *a):
        self._load()
        return super().pop(*a)

    def popitem(self, last=True):
        self._load()
        return super().popitem(last)

    def setdefault(self, k, default=None):
        self._load()
        return super().setdefault(k, default)

    def __iter__(self):       # keys are known up front, but dict(self) / OrderedDict(self) read values through the C fast path
        self._load()
        return super().__iter__()


_GC_FROZEN = [False]


def _freeze_gc_once():
    """Once the launch chains are recorded, move everything alive (torch, the model, the engine's buffers, the tapes: hundreds
    of thousands of long-lived container objects) out of the cyclic collector's working set.  A full (generation-2) collection
    over them takes tens of milliseconds - measured on the MI355X host: one such pause inside 20 ResNet-18 steps showed up as
    +3.5 ms per step although the GPU never waited for the host otherwise.  The objects stay alive anyway; young garbage is
    still collected.  Process-wide, therefore OPT-IN: VFS_GC_FREEZE=1 (bench.py sets it; a training script that owns its
    process can too)."""
    if _GC_FROZEN[0] or os.environ.get('VFS_GC_FREEZE', '0') != '1':
        return
    import gc
    gc.collect()
    gc.freeze()
    _GC_FROZEN[0] = True


class _GraphState:
    """Replay state of the fused step: after one eager step with a given input signature the forward chain and
    the backward chain are each recorded once - as a host-side command tape (default) or captured into a hipGraph
    (VFS_GRAPHS=1; torch.cuda.CUDAGraph over the launch stream) - and replayed afterwards; every kernel argument
    is a pointer into the engine's persistent buffers."""

    def __init__(self, key):
        self.key, self.warm = key, 0
        self.fwd = self.bwd = None
        self.imgs = self.loss = self.gl = self.ctx = self.means = None
        self.nbt = None
        self.borrowed = True      # read the caller's input buffer in place until it changes


class BaseTracker(nn.Module):
    """trackers/base.py:12-156."""

    def __init__(self, backbone, cls_head=None, train_cfg=None, test_cfg=None):
        super().__init__()
        self.backbone = builder.build_backbone(backbone)
        if cls_head is not None:
            self.cls_head = builder.build_head(cls_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights()
        self.fp16_enabled = False
        self.register_buffer('iteration', torch.tensor(0, dtype=torch.float))
        self._flat = None

    @property
    def with_cls_head(self):
        return hasattr(self, 'cls_head') and self.cls_head is not None

    def init_weights(self):
        self.backbone.init_weights()
        if self.with_cls_head:
            self.cls_head.init_weights()

    def extract_feat(self, imgs):
        return self.backbone(imgs)

    def forward_train(self, imgs, labels=None):
        raise NotImplementedError

    def forward_test(self, imgs, **kwargs):
        raise NotImplementedError

    @staticmethod
    def _parse_losses(losses):
        """base.py:76-110: mean every entry, sum the keys containing 'loss', average each
        log var over ranks.  The per-key all-reduces / .item() syncs of the reference are
        batched into one small all-reduce and one host read; values are identical."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        packed = torch.stack([v.detach().float().reshape(()) for v in log_vars.values()])
        if dist.is_available() and dist.is_initialized():
            packed = packed / dist.get_world_size()
            dist.all_reduce(packed)
        for k, v in zip(list(log_vars.keys()), packed.tolist()):
            log_vars[k] = v
        return loss, log_vars

    def forward(self, imgs, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(imgs, **kwargs)
        return self.forward_test(imgs, **kwargs)

    def train_step(self, data_batch, optimizer, **kwargs):
        """base.py:119-156; backward + optimizer step stay with the caller (OptimizerHook)."""
        self.iteration += 1
        losses = self(**data_batch)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(next(iter(data_batch.values()))))

    def val_step(self, data_batch, optimizer, **kwargs):
        losses = self(data_batch['imgs'], data_batch['ref_seg_map'], data_batch['img_meta'])
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(next(iter(data_batch.values()))))

    # ------------------------------------------------------------------ flat parameter arena
    def flatten_parameters(self):
        """Move every parameter (and its gradient) into one fp32 arena each, keeping the
        nn.Parameter objects and state_dict names: enables the one-launch SGD and bucketed
        gradient all-reduce.  Call after .to(device)."""
        params = [p for p in self.parameters()]
        if not params:
            return None
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
            p.grad = gflat[o:o + p.numel()].view(p.shape)
        self._flat = dict(params=flat, grads=gflat, offsets=offs, plist=params, device=dev)
        return self._flat

    def _ensure_arena(self):
        p0 = next(self.parameters())
        if self._flat is None or self._flat['device'] != p0.device or \
                self._flat['plist'][0].data_ptr() != self._flat['params'].data_ptr():
            self.flatten_parameters()
        f = self._flat
        if any(p.grad is None for p in f['plist']):     # zero_grad(set_to_none=True) happened
            f['grads'].zero_()
            for p, o in zip(f['plist'], f['offsets']):
                p.grad = f['grads'][o:o + p.numel()].view(p.shape)
        return f


@TRACKERS.register_module()
class SimSiamBaseTracker(BaseTracker):
    """trackers/sim_siam_base_tracker.py:8-79."""

    def __init__(self, *args, backbone, img_head=None, **kwargs):
        super().__init__(*args, backbone=backbone, **kwargs)
        if img_head is not None:
            self.img_head = builder.build_head(img_head)
        self.init_extra_weights()
        self.intra_video = False
        self.transpose_temporal = False
        if self.train_cfg is not None:
            self.intra_video = self.train_cfg.get('intra_video', False)
            self.transpose_temporal = self.train_cfg.get('transpose_temporal', False)
        self._anchor = None
        self._ctx = None
        self._works = []
        self._bf16_pending = []
        self.grad_bucket_bytes = 25 * 1024 * 1024

    @property
    def with_img_head(self):
        return hasattr(self, 'img_head') and self.img_head is not None

    def init_extra_weights(self):
        if self.with_img_head:
            self.img_head.init_weights()

    # ------------------------------------------------------------------ the HIP step
    def _chain_mode(self, dev):
        """how the forward / backward launch chains are issued after the first (eager) step:
        'tape'  - default: the chain's C-ABI calls and its stream waits / RCCL calls are recorded once as a host-side
                  command list (_lib.Tape) and replayed by a tight loop (~4 us of CPU per launch instead of ~25);
                  works with collectives (N > 1), and on MI355X it also beats the hipGraph replay of the same chain
                  (R50 10.7 -> 10.1 ms, R18 8.04 -> 7.81: the weight-gradient stream overlaps better with direct launches);
        'graph' - VFS_GRAPHS=1, single process on a GPU: each chain is captured once into a hipGraph and replayed;
        None    - plain eager (profiling, VFS_TAPE=0)."""
        eng = shared_engine()
        if eng.prof is not None:
            return None
        if dev.type == 'cuda' and os.environ.get('VFS_GRAPHS', '0') == '1' and not eng.collectives_on:
            return 'graph'
        return 'tape' if os.environ.get('VFS_TAPE', '1') == '1' else None

    def _step_forward(self, imgs):
        dev = imgs.device
        bump_params_epoch()      # the BatchNorm kernels update running_mean / running_var through raw pointers (eager or replayed)
        mode = self._chain_mode(dev)
        if mode is None:
            self._gs = None
            return self._hip_forward_train(imgs)
        f = self._ensure_arena()              # recorded chains hold raw pointers into the parameter / gradient arenas
        # recorded chains hold raw pointers into the engine's buffers: any (re)allocation there - another model attached
        # to the shared engine, a larger workspace, repacked weights - bumps eng.generation and forces a re-record
        key = (tuple(imgs.shape), imgs.dtype, dev, self.training, id(shared_engine()), shared_engine().generation, mode,
               f['params'].data_ptr(), f['grads'].data_ptr())
        gs = getattr(self, '_gs', None)
        if gs is None or gs.key != key:
            gs = self._gs = _GraphState(key)
            self.chain_resets = getattr(self, 'chain_resets', 0) + 1      # diagnostics: how often the recorded chains were dropped
        if gs.warm < 1:                       # first step eager: buffers, workspaces, packed weights settle
            gs.warm += 1
            return self._hip_forward_train(imgs)
        eng = shared_engine()
        self._ensure_arena()
        if gs.fwd is not None and gs.borrowed and imgs.data_ptr() != gs.imgs.data_ptr():
            if dev.type == 'cuda':
                torch.cuda.synchronize(dev)   # the old graphs may still be executing: never destroy them in flight
            gs.fwd = gs.bwd = None            # the caller moved on to another buffer: re-capture on a staging copy
            gs.borrowed = False
        recorded_now = False
        if gs.fwd is None:
            # a caller that keeps feeding the SAME resident fp32 buffer (bench.py, a device-side loader ring) is
            # read in place; anything else is staged into a private copy the captured chain reads
            inplace = gs.borrowed and imgs.dtype == torch.float32 and imgs.is_contiguous() and not imgs.requires_grad
            gs.imgs = imgs.detach() if inplace else imgs.detach().clone().contiguous().float()
            gs.borrowed = inplace
            before = {id(u): getattr(u, 'nbt_pending', 0) for u in eng.units}
            if mode == 'graph':
                torch.cuda.synchronize(dev)
                gs.fwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs.fwd):
                    gs.loss = self._hip_forward_train(gs.imgs)
            else:                             # the recording pass IS this step's forward
                tape = eng.begin_tape()
                try:
                    gs.loss = self._hip_forward_train(gs.imgs)
                finally:
                    eng.end_tape()
                gs.fwd = tape
                recorded_now = True
            gs.ctx = self._ctx
            gs.means = self._loss_means
            gs.nbt = [(u, getattr(u, 'nbt_pending', 0) - before[id(u)]) for u in eng.units]
            if mode == 'graph':
                for u, n in gs.nbt:           # the capture pass itself launched nothing
                    u.nbt_pending = before[id(u)]
        if not recorded_now:
            if not gs.borrowed:
                gs.imgs.copy_(imgs)
            gs.fwd.replay()
            for u, n in gs.nbt:
                if n:
                    u.nbt_pending = getattr(u, 'nbt_pending', 0) + n
        self._ctx = gs.ctx
        self._loss_means = gs.means
        return gs.loss

    def _step_backward(self, gl):
        gs = getattr(self, '_gs', None)
        if gs is None or gs.fwd is None or self._ctx is not gs.ctx:
            return self._hip_backward(gl)
        if gs.bwd is None:
            gs.gl = gl.detach().clone().contiguous().float()
            if isinstance(gs.fwd, torch.cuda.CUDAGraph):
                torch.cuda.synchronize(gl.device)
                gs.bwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs.bwd):
                    self._hip_backward(gs.gl)
            else:
                eng = shared_engine()
                tape = eng.begin_tape()
                try:
                    self._hip_backward(gs.gl)     # recording pass = this step's backward
                finally:
                    eng.end_tape()
                gs.bwd = tape
                _freeze_gc_once()
                return
        gs.gl.copy_(gl)
        gs.bwd.replay()
        self._ctx = None

    def _check_trainable(self):
        """the backbone's freezing options (frozen_stages / norm_eval / partial_bn, resnet.py:577-654) are on the HIP path: eval-mode
        BatchNorm backward without statistic terms, no weight gradients for frozen layers, propagation stops at the frozen
        prefix.  The head has no such option in the reference: a frozen / eval-mode head layer is refused rather than guessed."""
        for name, m in self.img_head.named_modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm) and not m.training:
                raise NotImplementedError(f'img_head.{name}: BatchNorm in eval mode inside forward_train is not on the HIP training path')
        for name, p in self.img_head.named_parameters():
            if not p.requires_grad:
                raise NotImplementedError(f'img_head.{name}: requires_grad=False inside forward_train is not on the HIP training path')
        bb = self.backbone
        frozen = [i for i, (_, m) in enumerate(bb.conv_modules()) if not m.conv.weight.requires_grad]
        if frozen and frozen != list(range(len(frozen))):
            raise NotImplementedError('frozen convolution weights must form a prefix of the backbone (frozen_stages); other patterns '
                                      'are not on the HIP training path')

    def _hip_forward_train(self, imgs):
        eng = shared_engine()
        dev = imgs.device
        self._check_trainable()
        self.backbone.attach(eng)
        self.img_head.attach(eng)
        self._ensure_arena()
        eng.p2p_chain_start(dev)
        eng.pack_weights()
        B, V, _, T, H, W = imgs.shape
        Nv = B * T
        N = V * Nv
        Wp = W + (W & 1)
        x4 = eng.buf('backbone.x4', (N, H, Wp, 4), BF16, dev)
        s = eng.stream(dev)
        eng.lib.imgs_to_nhwc4(imgs.contiguous().float(), x4, B, V, T, H, W, Wp, s)
        outs, bctx = self.backbone.forward_nhwc(eng, x4, N, H, W, V, True)
        last = max(outs)
        feat, h, w, C = outs[last]
        z, p, hctx = self.img_head.forward_nhwc(eng, feat, N, h, w, C, V, True)
        K = T if self.intra_video else 1
        weight = (1.0 / T if self.intra_video else 1.0) * float(self.img_head.loss_feat.loss_weight)
        neg = int(self.img_head.loss_feat.negative)
        loss = torch.empty(K, Nv, dtype=torch.float32, device=dev)
        eng.lib.cosine_loss_fwd(p[:Nv], z[:Nv], p[Nv:], z[Nv:], loss, Nv, p.shape[1], T, K, neg, weight, s)
        self._loss_means = eng.buf('img_head.loss_means', (K + 1,), torch.float32, dev)
        eng.lib.loss_means(loss, self._loss_means, K, Nv, s)       # what _parse_losses needs, inside the forward chain
        self._ctx = dict(bctx=bctx, hctx=hctx, z=z, p=p, Nv=Nv, T=T, K=K, weight=weight, neg=neg, last=last)
        return loss

    def _hip_backward(self, gl):
        eng = shared_engine()
        c = self._ctx
        if c is None:
            raise RuntimeError('backward called without a matching forward_train')
        z, p, Nv = c['z'], c['p'], c['Nv']
        dev = p.device
        s = eng.stream(dev)
        dp = eng.buf('img_head.dp', p.shape, BF16, dev)
        eng.p2p_chain_start(dev)
        gl = gl.contiguous().float()
        eng.lib.cosine_loss_bwd(p[:Nv], z[:Nv], p[Nv:], z[Nv:], gl, dp[:Nv], dp[Nv:], Nv, p.shape[1], c['T'],
                                c['K'], c['neg'], c['weight'], s)
        # split-K partials of the weight gradients are reduced by one table-driven launch per stage (data parallel: the
        # stage's gradients must be final before their all-reduce) or one for the whole step (single process)
        eng.defer_wgrad = os.environ.get('VFS_WGRAD_BATCH', '0') == '1'
        # Data parallel: the gradients of a stage are all-reduced as soon as they are final.  Round 6: the buckets are issued FROM THE
        # SIDE STREAM (the stream the weight gradients run on) - rounds 2-5 joined the side stream into the main chain at every stage
        # boundary, so the dgrad chain waited five times per step for the weight-gradient stream it is supposed to run ahead of.
        ddp_side = os.environ.get('VFS_DDP_SIDE', '1') == '1'

        def reduce_now(rng):
            if not eng.collectives_on:
                return
            eng.flush_wgrad(dev)
            if ddp_side and dev.type == 'cuda' and os.environ.get('VFS_SIDE_STREAM', '1') == '1':
                with eng.on_side_stream(dev):
                    self._allreduce_range(rng)
            else:
                eng.wgrad_join(dev)
                self._allreduce_range(rng)
        try:
            gfeat = self.img_head.backward_nhwc(eng, c['hctx'], dp)
            reduce_now(self._head_range())

            def stage_done(module):      # gradients of `module` are final: reduce them while earlier stages run
                reduce_now(self._param_range(module))
            self.backbone.backward_nhwc(eng, c['bctx'], {c['last']: gfeat}, on_stage_done=stage_done)
            eng.flush_wgrad(dev)
        finally:
            eng.defer_wgrad = False
        eng.wgrad_join(dev)
        x = eng._p2p
        if eng.collectives_on and x is not None and x.state.device == dev:
            # SyncBN window exchange: a rank whose wait timed out has NaN statistics, skips its own update (optim.py) - and has
            # just sent NaN gradients into the buckets above.  The error word is MAX-reduced over the ranks behind the last
            # bucket, so EVERY rank's sgd_step sees it and leaves weights and momentum untouched: the replicas stay identical
            # and the checkpoint writer never holds a poisoned step (round 4 advisor finding: the guard was local).
            eng.record(self._issue_allreduce, x.state[1:2], dist.ReduceOp.MAX)
        eng.record(self._wait_works)
        if self._bf16_pending:                  # bf16 buckets: back to the fp32 gradient arena once the collectives are done
            stage, g = eng.bufs['ddp.grad_bf16'], self._flat['grads']
            for a, b in self._bf16_pending:
                eng.lib.bf16_to_f32(stage[a:b], g[a:b], b - a, eng.stream(dev))
            del self._bf16_pending[:]
        self._ctx = None

    # ------------------------------------------------------------------ data-parallel gradients
    def _param_range(self, module):
        f = self._flat
        ids = {id(p) for p in module.parameters()}
        offs = [(o, p.numel()) for p, o in zip(f['plist'], f['offsets']) if id(p) in ids]
        if not offs:
            return (0, 0)
        return (min(o for o, _ in offs), max(o + (n + 3) // 4 * 4 for o, n in offs))

    def _head_range(self):
        return self._param_range(self.img_head)

    def _backbone_range(self):
        return self._param_range(self.backbone)

    def _allreduce_range(self, rng):
        """mean-all-reduce flat_grads[lo:hi] in ~25 MB buckets (torch DDP's default bucket size), asynchronously; the
        work handles are waited for by _wait_works at the end of the backward chain."""
        eng = shared_engine()
        if not eng.collectives_on:
            return
        lo, hi = rng
        g = self._flat['grads']
        cur = torch.cuda.current_stream(g.device) if g.device.type == 'cuda' else None      # the stream the buckets are ordered behind
        world = dist.get_world_size()
        step = max(1, self.grad_bucket_bytes // 4)
        bf16 = os.environ.get('VFS_GRAD_BF16', '0') == '1'      # opt-in: bf16 buckets halve the xGMI traffic (the reference reduces fp32)
        if bf16:
            stage = eng.buf('ddp.grad_bf16', (g.numel(),), BF16, g.device)
        # the collective library's own mean (every rank's contribution multiplied by 1 / world on its way into the sum - the very
        # products the separate `scale` launch made, so the bits do not change) saves one pass over the 153 MB gradient arena per step
        avg = (not bf16 and g.device.type == 'cuda' and dist.get_backend() == 'nccl' and os.environ.get('VFS_DDP_AVG', '1') == '1')
        for a in range(lo, hi, step):
            b = min(hi, a + step)
            chunk = g[a:b]
            if bf16:
                eng.lib.f32_to_bf16(chunk, stage[a:b], b - a, 1.0 / world, eng.stream(chunk.device))
                eng.record(self._issue_allreduce, stage[a:b], dist.ReduceOp.SUM, cur)
                self._bf16_pending.append((a, b))
                continue
            if avg:
                eng.record(self._issue_allreduce, chunk, dist.ReduceOp.AVG, cur)
                continue
            if chunk.device.type == 'cuda':
                eng.lib.scale(chunk, b - a, 1.0 / world, eng.stream(chunk.device))
            else:
                eng.record(chunk.mul_, 1.0 / world)
            eng.record(self._issue_allreduce, chunk, dist.ReduceOp.SUM, cur)

    def _issue_allreduce(self, chunk, op, stream=None):
        """asynchronous all-reduce behind `stream` (a recorded chain replays this call outside the stream context it was issued in)"""
        if stream is not None:
            with torch.cuda.stream(stream):
                w = dist.all_reduce(chunk, op=op, async_op=True)
        else:
            w = dist.all_reduce(chunk, op=op, async_op=True)
        if w is not None:
            self._works.append(w)

    def _wait_works(self):
        for w in self._works:
            w.wait()
        del self._works[:]

    # ------------------------------------------------------------------ reference API
    def train_step(self, data_batch, optimizer, **kwargs):
        """base.py:119-156.  Single process: forward_train + _parse_losses with the reduction done by
        vfs_loss_means inside the forward chain and ONE host read for the log vars, instead of ~30 tiny
        eager torch launches between the forward and the backward chain (they left the GPU idle for
        0.33 ms of an 8.5 ms ResNet-18 step).  Same keys, same values (double accumulation), same autograd
        contract: outputs['loss'].backward() runs the backward chain."""
        if not self.with_img_head or os.environ.get('VFS_FAST_TRAIN_STEP', '1') != '1' or \
                not {'imgs'} <= set(k for k, v in data_batch.items() if v is not None) <= {'imgs', 'label'}:
            return super().train_step(data_batch, optimizer, **kwargs)
        self.iteration += 1
        imgs = data_batch['imgs']
        if self.transpose_temporal:
            imgs = imgs.transpose(1, 3).contiguous()
        assert imgs.size(1) == 2
        assert imgs.ndim == 6
        if self._anchor is None or self._anchor.device != imgs.device:
            self._anchor = torch.zeros(1, device=imgs.device, requires_grad=True)
        rows = _TrainStepFn.apply(self._anchor, self, imgs)
        loss = _ReduceLossFn.apply(rows, self._loss_means)
        if dist.is_available() and dist.is_initialized():      # log vars are averaged over the ranks (base.py:103-108)
            packed = self._loss_means / dist.get_world_size()
            dist.all_reduce(packed)
        else:
            packed = self._loss_means.clone()                  # the engine buffer is rewritten by the next step
        keys = [f'img_head.{i}.loss_feat' for i in range(rows.shape[0])] + ['loss']
        if os.environ.get('VFS_LAZY_LOG', '1') == '1':
            if packed.is_cuda and os.environ.get('VFS_LOG_ASYNC', '1') == '1':
                # the values travel to pinned host memory right behind the forward chain; reading them later waits for THAT
                # copy only.  A blocking .tolist() on the device tensor drains the whole stream instead: a consumer that
                # looks once per iteration (mmcv's log buffer, bench.py) then restarts the launch queue from empty every
                # step - ~0.1 ms of GPU idle time at each step boundary.
                host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
                host.copy_(packed, non_blocking=True)
                x = shared_engine()._p2p      # SyncBN window exchange: its error word travels with the log values
                if x is not None:
                    xflag = torch.empty(1, dtype=torch.int64, pin_memory=True)
                    xflag.copy_(x.state[1:2], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                packed = (host, ev) if x is None else (host, ev, xflag)
            log_vars = LazyLogVars(keys, packed)
        else:
            log_vars = OrderedDict(zip(keys, packed.tolist()))
            x = shared_engine()._p2p
            if x is not None:                 # the read above drained the forward chain: its exchanges have run
                x.raise_if_failed()
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data_batch['imgs']))

    def forward_train(self, imgs, grids=None, label=None):
        """imgs [B,2,3,T,H,W] -> {'img_head.{i}.loss_feat': [B*T]} (sim_siam_base_tracker.py:58-76)."""
        if self.transpose_temporal:
            imgs = imgs.transpose(1, 3).contiguous()
        assert imgs.size(1) == 2
        assert imgs.ndim == 6
        if not self.with_img_head:
            return dict()
        if self._anchor is None or self._anchor.device != imgs.device:
            self._anchor = torch.zeros(1, device=imgs.device, requires_grad=True)
        loss = _TrainStepFn.apply(self._anchor, self, imgs)
        losses = {f'{i}.loss_feat': loss[i] for i in range(loss.shape[0])}
        return add_prefix(losses, prefix='img_head')

    def forward_test(self, imgs, **kwargs):
        raise NotImplementedError


@TRACKERS.register_module()
class VanillaTracker(BaseTracker):
    """trackers/vanilla_tracker.py:16-206: DAVIS label propagation."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.save_np = self.test_cfg.get('save_np', False)

    @property
    def stride(self):
        assert isinstance(self.backbone, ResNet)
        end = self.backbone.original_out_indices[0]
        return int(np.prod(self.backbone.strides[:end + 1]) * 4)

    def forward_train(self, imgs, labels=None):
        raise NotImplementedError

    def forward_test(self, imgs, ref_seg_map, img_meta):
        from .labelprop import forward_test_hip
        return forward_test_hip(self, imgs, ref_seg_map, img_meta)
