"""Host-side executor: sequences the libvfs_hip.so kernels for ResNet / SimSiam-head forward
and backward over NHWC bf16 buffers.  PyTorch supplies device memory (torch.empty), streams and
torch.distributed; every number on the path is produced by the HIP kernels.

Layout of one "unit" (mmcv ConvModule = conv -> BN -> act in the reference):
    raw = conv(a_prev)                 bf16, BatchNorm partial statistics from the GEMM epilogue
    bnp = finalize(statistics)         float[G][4][C]   (G independent BN batches = views)
    act = relu(raw*scale+shift [+res]) bf16
Backward of a unit: bn_bwd_reduce -> (all-reduce) -> bn_bwd_apply -> wgrad (+dgrad).
"""
import math
import os

import torch
import torch.distributed as dist

from ._lib import Tape, TapeLib, get_lib
from .packing import (igemm_ksplit, bn_fold_eligible, build_pack_table, build_reduce_table, conv_halo_eligible, conv_stats_rows, wgrad_halo_eligible,
                      wgrad_inl_floats, wgrad_splits, small_map)

BF16 = torch.bfloat16

FIN_FUSE = os.environ.get('VFS_FIN_FUSE', '1') == '1'      # BatchNorm statistics finished in the apply kernels' prologue
COARSE_ROWS = int(os.environ.get('VFS_COARSE_ROWS', '0'))      # conv launches sum their statistics rows in groups (vfs_conv_fwd_coarse): 0 off, 1 implicit-GEMM kernels, 2 halo kernels too
TILES_PER_TICKET = [1 << 16]      # capacity (uint32 words) of the shared ticket buffer
FIN_MAX_ROWS = int(os.environ.get('VFS_FIN_MAX_ROWS', '128'))   # ... for at most this many statistics rows per group
FIN_XCHG = os.environ.get('VFS_FIN_XCHG', '0') == '1'      # SyncBN (round 6, opt-in): ... and the window exchange folded into the same launch (vfs_bn_act_fin_xchg / vfs_bn_bwd_apply_fin_xchg) - measured LEVEL with the reduction + exchange launch of rounds 3-5 on one GPU (8.92 vs 8.88 ms), and its waiting workgroups hold their CUs, so the default stays 0
# timing experiments only: kernel families (Engine.timed labels) whose launches are dropped - results are garbage, the step
# time shows what the family costs on the critical path (no kernel here has data-dependent control flow)
MASK_ADD = os.environ.get('VFS_MASK_ADD', '1') == '1'     # with MASK_BITS: the identity-branch gradient g * (y > 0) is applied on the fly (vfs_conv_dgrad_maskadd), never written
MASK_BITS = os.environ.get('VFS_MASK_BITS', '1') == '1'   # residual joins also write a bit-packed ReLU mask; their BatchNorm backward reads it instead of y (R50 -0.3 ms)
NOMASK = os.environ.get('VFS_DEBUG_NOMASK') == '1'     # what-if timing: BatchNorm backward without reading the activation as ReLU mask
SKIP = frozenset(filter(None, os.environ.get('VFS_DEBUG_SKIP', '').split(',')))
KSPLIT = os.environ.get('VFS_KSPLIT', '0') == '1'     # split-K for the head's Linear layers (slower as measured)
WGRAD_INL = os.environ.get('VFS_WGRAD_INL', '0') == '1'      # round 6, opt-in: split-K reduction of the weight gradients inside the launch (vfs_conv_wgrad_inl) - measured SLOWER (R50 7.97 -> 10.8 ms: device-scope sc1 accesses move ~0.2 TB/s, MEASUREMENTS.md); default: kernel + wgrad_reduce


class ConvUnit:
    """Host state of one conv / linear layer (+ its BatchNorm)."""

    def __init__(self, name, weight, bias, bn, k, stride, pad, kind='conv', dil=1):
        self.name, self.weight, self.bias, self.bn = name, weight, bias, bn
        self.k, self.stride, self.pad, self.kind, self.dil = k, stride, pad, kind, dil
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.wf = self.wd = None
        self.bnp = self.sums = self.bsums = None
        self.need_wd = True

    def out_hw(self, H, W):
        span = self.dil * (self.k - 1) + 1
        return (H + 2 * self.pad - span) // self.stride + 1, (W + 2 * self.pad - span) // self.stride + 1


class Engine:
    def __init__(self, lib=None):
        self.lib = lib if lib is not None else get_lib()
        self.n_wgrad_tickets = int(self.lib.cfunc('wgrad_tickets')())      # size of the ticket array of vfs_conv_wgrad_inl
        self.bufs = {}
        self.units = []
        self._pack = None
        self._pack_key = None
        self.process_group = None
        self._side = {}
        self._side_dirty = False
        self.prof = None     # list collecting (kind, flops, start_event, end_event) when profiling
        self.tape = None
        # weight-gradient split-K partials: reduced per layer right after the kernel (default), or - inside the trackers'
        # backward chain (defer_wgrad) - kept in per-layer buffers and reduced by ONE table-driven launch per flush
        self.defer_wgrad = False
        self._wpending = []
        self._wtables = {}
        self._p2p, self._p2p_tried = None, False      # SyncBN statistic exchange over xGMI (vfs_amd/p2p.py), set up lazily
        self.generation = 0  # bumped whenever a persistent buffer is (re)allocated: recorded launch chains hold raw pointers
        for kv in filter(None, os.environ.get('VFS_OPTS', '').split(',')):      # kernel A/B knobs: "name=value,..."
            name, value = kv.split('=')
            self.lib.set_option(name.strip().encode(), int(value))
            if name.strip() == 'halo_min_fill':      # the host mirrors of the tiling rules must agree with the library
                from . import packing
                packing.HALO_MIN_FILL = int(value)

    # ------------------------------------------------------------------ command tape (see _lib.Tape)
    def begin_tape(self):
        assert self.tape is None
        self.tape = Tape(self.lib)
        self._real_lib, self.lib = self.lib, TapeLib(self.lib, self.tape)
        return self.tape

    def end_tape(self):
        self.lib, self.tape = self._real_lib, None

    def record(self, fn, *args):
        """run a Python-side action of the chain (stream wait, collective) and put it on the tape if one is recording"""
        if self.tape is not None:
            self.tape.ops.append((None, fn, args))
        return fn(*args)

    # ------------------------------------------------------------------ plumbing
    @property
    def world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.process_group)
        return 1

    @staticmethod
    def stream(dev):
        if dev.type == 'cuda':
            return torch.cuda.current_stream(dev).cuda_stream
        return None

    def buf(self, key, shape, dtype, dev):
        t = self.bufs.get(key)
        shape = tuple(int(s) for s in shape)
        if t is None or t.shape != shape or t.dtype != dtype or t.device != dev:
            if t is not None and dev.type == 'cuda':
                torch.cuda.synchronize(dev)     # a recorded chain may still be reading the old block
            t = torch.empty(shape, dtype=dtype, device=dev)
            self.bufs[key] = t
            self.generation += 1
        return t

    def ws(self, key, numel, dtype, dev):
        """grow-only flat workspace"""
        t = self.bufs.get(key)
        if t is None or t.numel() < numel or t.dtype != dtype or t.device != dev:
            if t is not None and dev.type == 'cuda':
                torch.cuda.synchronize(dev)     # the old block may still be in use on the side stream
            t = torch.empty(int(numel), dtype=dtype, device=dev)
            self.bufs[key] = t
            self.generation += 1
        return t

    @staticmethod
    def conv_kind(u, N, H, W, dgrad=False):
        """kernel family a forward / dgrad launch of unit u lands in (bench.py's per-kernel roofline classes; mirrors the
        dispatch of csrc/conv_igemm.hip): stem_fwd | conv3x3_halo | conv_igemm"""
        if u.kind == 'stem':
            return 'stem_fwd'
        cin, cout = (u.cout, u.cin) if dgrad else (u.cin, u.cout)
        return 'conv3x3_halo' if (u.dil == 1 and conv_halo_eligible(N, H, W, cin, cout, u.k, u.stride, u.pad)) else 'conv_igemm'

    def timed(self, kind, work, dev, fn, *args):
        """launch through `fn`; with profiling on, bracket it with events on the launch stream.
        work = algorithmic FLOP of the launch, or (FLOP, algorithmic HBM bytes: every operand once)"""
        flops, nbytes = work if isinstance(work, tuple) else (work, 0.0)
        if SKIP and kind in SKIP:      # what-if measurement (VFS_DEBUG_SKIP=family,...): the step WITHOUT this family's launches
            return 0
        if self.prof is None or dev.type != 'cuda':
            if self.tape is not None:      # recording: the launch carries its family and algorithmic work onto the tape (_lib.Tape.meta)
                self.tape.next_meta = (kind, flops, nbytes)
                try:
                    return fn(*args)
                finally:
                    self.tape.next_meta = None
            return fn(*args)
        pool = getattr(self, 'prof_pool', None)
        if pool:                       # pre-created events: creation is the expensive part
            e0, e1 = pool.pop(), pool.pop()
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        self.prof.append((kind, flops, e0, e1, nbytes))
        return rc

    @property
    def collectives_on(self):
        """collectives run when world > 1; VFS_FORCE_COLLECTIVES=1 also runs them in a 1-rank group
        (exercises the RCCL calls, dtypes and stream ordering on a single-GPU box)"""
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return self.world > 1 or os.environ.get('VFS_FORCE_COLLECTIVES') == '1'

    def allreduce(self, t):
        """sum a SyncBN statistic buffer over the ranks: the P2P window exchange (vfs_amd/p2p.py; one small kernel, on the launch
        stream and on a recording tape like any other C-ABI call) when it is up, a collective-library all-reduce otherwise"""
        if not self.collectives_on:
            return
        x = self.p2p_exchange(t.device)
        if x is not None and x.fits(t):
            x.allreduce(self.lib, t, self.stream(t.device))
        else:
            self.record(self._all_reduce, t)

    def p2p_exchange(self, dev):
        """the SyncBN exchange of this process, set up on first use (VFS_SYNCBN_P2P=0: never; GPUs only; world > 1 - a 1-rank
        group with VFS_FORCE_COLLECTIVES=1 keeps exercising the collective-library calls).  Set-up or self-test failure on any
        rank -> None for good (collective library)."""
        if self._p2p is not None or self._p2p_tried:
            return self._p2p
        self._p2p_tried = True
        want = os.environ.get('VFS_SYNCBN_P2P', '1')      # 'force': also in a 1-rank group (measures the exchange kernels' cost on one GPU)
        if want not in ('1', 'force') or dev.type != 'cuda' or (self.world < 2 and want != 'force'):
            return None
        from .p2p import P2PExchange
        x, ok = None, 1
        try:
            x = P2PExchange(self._real_lib if self.tape is not None else self.lib, dev, self.process_group)
        except Exception as e:      # noqa: BLE001
            print(f'[vfs_amd] SyncBN P2P exchange unavailable ({type(e).__name__}: {e}); using the collective library', flush=True)
            ok = 0
        # every rank must take the same path: agree before (and inside) the self-test
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if dist.get_backend(self.process_group) == 'nccl' else 'cpu')
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.process_group)
        if int(flag.item()) and x.self_test(self.stream(dev)):
            self._p2p = x
            self.generation += 1
        elif x is not None:
            if ok:
                print('[vfs_amd] SyncBN P2P exchange failed its self-test; using the collective library', flush=True)
            x.close()
        return self._p2p

    def p2p_chain_start(self, dev):
        """head of a launch chain (forward / backward of a train step): the folded SyncBN exchanges of the chain are numbered from 0
        and the chain counter in device memory moves on (vfs_p2p_chain_start; recorded on the tape like any other call)"""
        self._xseq = 0
        if not (self.collectives_on and FIN_XCHG and FIN_FUSE):
            return
        x = self.p2p_exchange(dev)
        if x is not None:
            self.lib.p2p_chain_start(x.state, self.stream(dev))

    def next_xseq(self):
        n = getattr(self, '_xseq', 0)
        self._xseq = n + 1
        return n

    def _all_reduce(self, t):
        dist.all_reduce(t, group=self.process_group)

    # ------------------------------------------------------------------ weights
    def register(self, unit):
        self.units.append(unit)
        self._pack = None
        self.generation += 1      # the pack table (and its bf16 weight copies) will be rebuilt
        return unit

    def pack_weights(self):
        """fp32 OIHW master weights -> bf16 MFMA layouts, all layers in ONE launch."""
        if not self.units:
            return
        dev = self.units[0].weight.device
        key = tuple((u.weight.data_ptr(), u.weight.device) for u in self.units)
        if self._pack is None or self._pack_key != key:
            entries = []
            for u in self.units:
                if u.kind == 'stem':
                    u.wf = torch.zeros(u.cout, 8, 8, 4, dtype=BF16, device=dev)
                    u.wd = None
                else:
                    u.wf = torch.empty(u.cout, u.k, u.k, u.cin, dtype=BF16, device=dev)
                    u.wd = torch.empty(u.cin, u.k, u.k, u.cout, dtype=BF16, device=dev) if u.need_wd else None
                entries.append((u.weight.data, u.wf, u.wd, 1 if u.kind == 'stem' else 0))
            if dev.type == 'cuda':
                torch.cuda.synchronize(dev)     # the previous packed copies may still be in use
            self._pack = build_pack_table(entries, dev)
            self._pack_key = key
            self.generation += 1
        tab, n, total = self._pack
        nel = sum(u.weight.numel() for u in self.units)
        self.timed('pack_weights', (0.0, 4.0 * nel + 2.0 * sum(u.weight.numel() * (2 if u.wd is not None else 1) for u in self.units)), dev,
                   self.lib.pack_weights, tab, n, total, self.stream(dev))

    # ------------------------------------------------------------------ forward primitives
    def conv_fwd(self, u, x, N, H, W, G, train, tag='', in_bn=None, defer_fin=False):
        """raw = conv(x); BN statistics/params when the unit has a BN.  Returns (raw, Ho, Wo)."""
        dev = x.device
        s = self.stream(dev)
        lib = self.lib
        if u.kind == 'stem':
            Ho, Wo = (H + 6 - 7) // 2 + 1, (u.true_w + 6 - 7) // 2 + 1
        else:
            Ho, Wo = u.out_hw(H, W)
        M = N * Ho * Wo
        y = self.buf(f'{u.name}{tag}.raw', (N, Ho, Wo, u.cout), BF16, dev)
        want_stats = u.bn is not None and train
        Ng = N // G
        mpg = Ng * Ho * Wo
        # statistics rows per group when ONE launch covers all groups (spatial tiles of the halo kernels, ragged
        # edges included, or linear 128-pixel blocks), else one launch per group
        rows = None if u.kind == 'stem' else conv_stats_rows(N, G, H, W, u.cin, u.cout, u.k, u.stride, u.pad, Ho, Wo,
                                                             halo=u.dil == 1)
        fused = rows is not None
        nblk_g = rows if fused else (mpg + 127) // 128
        if u.kind == 'stem':      # the stem kernel emits one statistics row per 8x16 spatial tile
            fused = True
            nblk_g = Ng * ((Ho + 7) // 8) * ((Wo + 15) // 16)
        # small groups that are not multiples of the 128-pixel statistics rows (the head's Linear layers on the ResNet-50
        # config: 32 rows per view): ONE launch without statistics rows, the sums come from the stored output
        raw_stats = (want_stats and not fused and not self.collectives_on and mpg <= 2048
                     and os.environ.get('VFS_RAW_STATS', '1') == '1')
        # round 6: Linear + BatchNorm1d + ReLU of the head in ONE launch (vfs_linear_bn_act: the workgroup that owns 16 output channels
        # owns all <= 64 rows, so the batch statistics are local to it) instead of conv + statistics + apply
        lin_fused = (raw_stats and u.kind == 'linear' and defer_fin and M <= 256 and G <= 4 and u.cin % 128 == 0 and u.cout % 16 == 0
                     and os.environ.get('VFS_HEAD_FUSE', '1') == '1')
        if raw_stats:
            fused, want_rows = True, False
        else:
            want_rows = want_stats
        partial = self.ws('ws.stats', G * nblk_g * 2 * u.cout, torch.float32, dev) if want_rows else None
        # large maps (more than FIN_MAX_ROWS statistics rows per group): the conv launch sums its rows in groups of 2^L itself, so
        # that the bn_act behind it can finish the statistics in its prologue and the reduction launch in between disappears
        coarse_l2, coarse_rows = 0, None
        if (COARSE_ROWS and want_rows and fused and defer_fin and FIN_FUSE and train and not self.collectives_on and in_bn is None
                and u.kind != 'stem' and u.dil == 1 and nblk_g > FIN_MAX_ROWS and (COARSE_ROWS >= 2 or self.conv_kind(u, N, H, W) == 'conv_igemm')):
            L = 1
            while (nblk_g >> L) > FIN_MAX_ROWS:
                L += 1
            if nblk_g % (1 << L) == 0 and L <= 8 and G * (nblk_g >> L) * ((u.cout + 63) // 64) <= TILES_PER_TICKET[0]:
                coarse_l2 = L
                coarse_rows = self.ws('ws.stats_coarse', G * (nblk_g >> L) * 2 * u.cout, torch.float32, dev)
        bias = u.bias.data if u.bias is not None else None
        groups = [(0, N, partial)] if fused else [
            (g * Ng, Ng, partial[g * nblk_g * 2 * u.cout:] if want_rows else None) for g in range(G)]
        if lin_fused:
            bn = u.bn
            u.sums = self.buf(f'{u.name}.sums', (G, 2, u.cout), torch.float64, dev)
            u.bnp = self.buf(f'{u.name}.bnp', (G, 4, u.cout), torch.float32, dev)
            act = self.buf(f'{u.name}{tag}.act', (N, u.cout), BF16, dev)      # (the shape bn_act gives the head: raw.view(N, cout))
            self.timed('conv_igemm', (2.0 * M * u.cout * u.cin, 2.0 * (M * u.cin + 2 * M * u.cout + u.cout * u.cin)), dev, lib.linear_bn_act,
                       x, u.wf, bias, bn.weight.data, bn.bias.data, y, act, u.bnp, u.sums, bn.running_mean, bn.running_var, M, u.cin, u.cout, mpg,
                       1 if getattr(u, 'relu', False) else 0, float(mpg), float(bn.eps), float(bn.momentum), s)
            u.nbt_pending = getattr(u, 'nbt_pending', 0) + G
            self._fused_act = (u, act)      # the bn_act call that follows hands this tensor out
            return y, Ho, Wo
        for n0, nn_, part in groups:
            if u.kind == 'stem':
                self.timed('stem_fwd', (2.0 * nn_ * Ho * Wo * 64 * 147, 2.0 * nn_ * (H * W * 4 + Ho * Wo * 64)), dev, lib.stem_fwd,
                           x[n0:n0 + nn_], u.wf, y[n0:n0 + nn_], part, nn_, H, W, Ho, Wo, s)
            elif in_bn is not None:     # x is the producer's RAW output: BatchNorm + ReLU folded into the operand load
                assert (n0, nn_) == (0, N), 'folded input BatchNorm needs the single-launch (fused statistics) path'
                self.timed('conv3x3_halo' if u.k == 3 else 'conv_igemm', (2.0 * nn_ * Ho * Wo * u.cout * u.k * u.k * u.cin,
                                            2.0 * (nn_ * H * W * u.cin + nn_ * Ho * Wo * u.cout + u.cout * u.k * u.k * u.cin)), dev, lib.conv_fwd_bnin,
                           x, in_bn[0], in_bn[1], u.wf, y, bias, part, nn_, H, W, u.cin, Ho, Wo, u.cout,
                           u.k, u.k, u.stride, u.pad, s)
            elif u.dil != 1:            # dilated taps: implicit-GEMM forward only (frozen backbones)
                self.timed('conv_igemm', (2.0 * nn_ * Ho * Wo * u.cout * u.k * u.k * u.cin,
                                          2.0 * (nn_ * H * W * u.cin + nn_ * Ho * Wo * u.cout + u.cout * u.k * u.k * u.cin)), dev,
                           lib.conv_fwd_dilated, x[n0:n0 + nn_], u.wf, y[n0:n0 + nn_], bias, part, nn_, H, W, u.cin, Ho, Wo, u.cout,
                           u.k, u.k, u.stride, u.pad, u.dil, s)
            else:
                ks, ksws = igemm_ksplit(nn_ * Ho * Wo, u.cout, u.k * u.k * u.cin) if (u.k == 1 and KSPLIT) else (1, 0)
                work = (2.0 * nn_ * Ho * Wo * u.cout * u.k * u.k * u.cin,
                        2.0 * (nn_ * H * W * u.cin + nn_ * Ho * Wo * u.cout + u.cout * u.k * u.k * u.cin))
                if ks > 1:      # few pixels, long reduction (the head's Linear layers): split-K fills the chip
                    self.timed('conv_igemm', work, dev, lib.conv_fwd_splitk, x[n0:n0 + nn_], u.wf, y[n0:n0 + nn_], bias, part,
                               self.ksplit_ws(ksws, dev), ks, nn_, H, W, u.cin, Ho, Wo, u.cout, u.k, u.k, u.stride, u.pad, s)
                elif coarse_l2:     # ... and the statistics rows summed in groups by the launch itself (vfs_conv_fwd_coarse)
                    self.timed(self.conv_kind(u, nn_, H, W), work, dev, lib.conv_fwd_coarse, x[n0:n0 + nn_], u.wf, y[n0:n0 + nn_], bias, part,
                               coarse_rows, self.stats_tickets(dev), coarse_l2, nn_, H, W, u.cin, Ho, Wo, u.cout, u.k, u.k, u.stride, u.pad, s)
                else:
                    self.timed(self.conv_kind(u, nn_, H, W), work, dev, lib.conv_fwd, x[n0:n0 + nn_], u.wf, y[n0:n0 + nn_], bias, part,
                               nn_, H, W, u.cin, Ho, Wo, u.cout, u.k, u.k, u.stride, u.pad, s)
        if u.bn is not None:
            bn = u.bn
            assert getattr(self, '_pending_fin', None) is None, 'a deferred BatchNorm finalisation was never consumed'
            if train:
                u.sums = self.buf(f'{u.name}.sums', (G, 2, u.cout), torch.float64, dev)
                u.bnp = self.buf(f'{u.name}.bnp', (G, 4, u.cout), torch.float32, dev)
                if coarse_l2:
                    self._pending_fin = (u, coarse_rows, nblk_g >> coarse_l2, float(mpg))
                elif (defer_fin and FIN_FUSE and fused and not raw_stats and not self.collectives_on and u.kind != 'stem'
                        and nblk_g <= FIN_MAX_ROWS):
                    # the caller's next launch is bn_act on this output: it finishes the statistics in its prologue
                    self._pending_fin = (u, partial, nblk_g, float(mpg))
                elif raw_stats:
                    self.timed('bn_stats', (0.0, 2.0 * M * u.cout), dev, lib.bn_stats_raw_finalize, y, u.sums, bn.weight.data, bn.bias.data, u.bnp, bn.running_mean, bn.running_var,
                                              G, mpg, u.cout, float(mpg), float(bn.eps), float(bn.momentum), s)
                elif self.collectives_on:     # SyncBN: statistics are summed over the ranks between the two stages
                    x = self.p2p_exchange(dev)
                    if (x is not None and x.fits(u.sums) and FIN_XCHG and defer_fin and FIN_FUSE and fused and u.kind != 'stem' and nblk_g <= FIN_MAX_ROWS
                            and G * 2 * min(u.cout, 64) <= 256 and u.cout <= 4096):
                        # few rows: the bn_act behind this launch sums them in its prologue AND runs the window exchange there
                        self._pending_fin = (u, partial, nblk_g, float(mpg * self.world), x)
                    elif x is not None and x.fits(u.sums):      # rows -> sums -> window exchange, one launch
                        self.timed('bn_stats', (0.0, 8.0 * G * nblk_g * u.cout), dev, lib.bn_reduce_partials_xchg, partial, u.sums,
                                   self.bn_scratch(G, u.cout, dev), G, nblk_g, u.cout, *x.tail_args(), s)
                    else:
                        self.timed('bn_stats', (0.0, 8.0 * G * nblk_g * u.cout), dev, lib.bn_reduce_partials, partial, u.sums,
                                   self.bn_scratch(G, u.cout, dev), G, nblk_g, u.cout, s)
                        self.allreduce(u.sums)
                    if getattr(self, '_pending_fin', None) is not None:
                        pass
                    elif defer_fin and FIN_FUSE and u.kind != 'stem':
                        # the bn_act that follows turns the all-reduced sums into scale / shift itself (any size)
                        self._pending_fin = (u, None, 0, float(mpg * self.world))
                    else:
                        lib.bn_finalize(u.sums, bn.weight.data, bn.bias.data, u.bnp, bn.running_mean, bn.running_var, G,
                                        u.cout, float(mpg * self.world), float(bn.eps), float(bn.momentum), s)
                else:
                    self.timed('bn_stats', (0.0, 8.0 * G * nblk_g * u.cout), dev, lib.bn_stats_finalize, partial, u.sums,
                               self.bn_scratch(G, u.cout, dev), bn.weight.data, bn.bias.data,
                                          u.bnp, bn.running_mean, bn.running_var, G, nblk_g, u.cout, float(mpg),
                                          float(bn.eps), float(bn.momentum), s)
                u.nbt_pending = getattr(u, 'nbt_pending', 0) + G   # num_batches_tracked, flushed lazily
            else:
                u.bnp = self.buf(f'{u.name}.bnp_eval', (1, 4, u.cout), torch.float32, dev)
                lib.bn_eval_params(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, u.bnp, u.cout,
                                   float(bn.eps), s)
        return y, Ho, Wo

    def can_fold_input_bn(self, u, N, G, H, W, train):
        """may conv unit u read the raw output of its producer (BatchNorm + ReLU folded into the load)?"""
        if os.environ.get('VFS_BNACT_FUSE', '1') != '1' or u.kind == 'stem' or u.dil != 1:
            return False
        # pays only where the saved activation pass is large: the fold costs VALU work in the staging of
        # the consumer's forward and weight-gradient kernels (round-2 whole-step A/B on MI355X, threshold 48 / 32 / 16 MB:
        # ResNet-50 9.36 / 9.30 / 9.24 ms, ResNet-18 unchanged - its folded tensors are all >= 48 MB)
        # whole-image tiles (maps of at most 8x8): the fold is a loss (ResNet-18 layer4 at 256^2, 17 MB: step 6.90 with, 6.82 without -
        # profiles/r06_fold3x3_threshold_step_ab.txt); VFS_BNACT_FUSE_SMALL=1 keeps it (tests exercise the kernels through the engine)
        if u.k == 3 and small_map(H, W) and os.environ.get('VFS_BNACT_FUSE_SMALL', '0') != '1':
            return False
        mb = os.environ.get('VFS_BNACT_FUSE_1X1_MB', '16') if u.k == 1 else os.environ.get('VFS_BNACT_FUSE_MB', '16')
        if N * H * W * u.cin * 2 < float(mb) * (1 << 20):
            return False
        Ng = N // G
        mpg = Ng * H * W
        if not (G == 1 or mpg % 128 == 0):      # the consumer must take the single-launch statistics path
            return False
        return bn_fold_eligible(N, G if train else 1, H, W, u.cin, u.cout, u.k, u.stride, u.pad)

    def ksplit_ws(self, need, dev):
        """workspace of the split-K kernels: 1024 ticket words (zero between launches) + partial tiles"""
        t = self.bufs.get('ws.ksplit')
        if t is None or t.numel() < need or t.device != dev:
            if t is not None and dev.type == 'cuda':
                torch.cuda.synchronize(dev)
            t = torch.zeros(int(need), dtype=torch.float32, device=dev)
            self.bufs['ws.ksplit'] = t
            self.generation += 1
        return t

    def bn_scratch(self, G, C, dev):
        """scratch of the chunked BatchNorm reductions: 64 ticket counters (must start at zero; every
        launch leaves them at zero) followed by double[G][128][2][C] chunk sums"""
        need = 32 + G * 128 * 2 * C
        t = self.bufs.get('ws.bnred')
        if t is None or t.numel() < need or t.device != dev:
            t = torch.zeros(need, dtype=torch.float64, device=dev)
            self.bufs['ws.bnred'] = t
            self.generation += 1
        return t

    def bn_act(self, u, raw, M, G, train, relu, res=None, rres=None, rbnp=None, tag='', want_mask=False):
        """y = [relu](bn(raw) [+ res] [+ bn(rres)]).  want_mask (the join of a residual block, training): also write the
        bit-packed mask y > 0 (uint8 [M][C/8], left in u.mask_bits) - the unit's BatchNorm backward reads it instead of y"""
        dev = raw.device
        fa = getattr(self, '_fused_act', None)
        if fa is not None:      # Linear + BatchNorm1d + ReLU ran as one launch (conv_fwd, lin_fused)
            self._fused_act = None
            assert fa[0] is u and res is None and rres is None and not want_mask and bool(relu) == bool(getattr(u, 'relu', False))
            u.mask_bits = None
            return fa[1]
        y = self.buf(f'{u.name}{tag}.act', raw.shape, BF16, dev)
        u.mask_bits = self.buf(f'{u.name}{tag}.mbits', (M * u.cout // 8,), torch.uint8, dev) if (want_mask and MASK_BITS and u.cout % 8 == 0 and (u.cout < 64 or u.cout % 64 == 0)) else None      # slab-major layout (mask8_index): whole 64-channel slabs
        mpg = M // G if train else M
        nbytes = 2.0 * M * u.cout * (2 + (res is not None) + (rres is not None)) + (M * u.cout / 8.0 if u.mask_bits is not None else 0.0)      # raw in, activation out, identity in (+ mask bits out)
        fin = getattr(self, '_pending_fin', None)
        if fin is not None:
            assert fin[0] is u, 'deferred BatchNorm finalisation belongs to another unit'
            self._pending_fin = None
            bn = u.bn
            if len(fin) > 4:      # SyncBN: statistics rows summed AND exchanged with the peers inside this launch
                self.timed('bn_act', (0.0, nbytes), dev, self.lib.bn_act_fin_xchg, raw, fin[1], fin[2], bn.weight.data, bn.bias.data, u.bnp, u.sums, bn.running_mean,
                           bn.running_var, res, rres, rbnp, y, u.mask_bits, M, u.cout, mpg, 1 if relu else 0, fin[3], float(bn.eps), float(bn.momentum),
                           *fin[4].tail_args(), self.next_xseq(), self.stream(dev))
                return y
            self.timed('bn_act', (0.0, nbytes), dev, self.lib.bn_act_fin_mask, raw, fin[1], fin[2], bn.weight.data, bn.bias.data, u.bnp, u.sums, bn.running_mean, bn.running_var,
                                res, rres, rbnp, y, u.mask_bits, M, u.cout, mpg, 1 if relu else 0, fin[3], float(bn.eps), float(bn.momentum),
                                self.stream(dev))
            return y
        self.timed('bn_act', (0.0, nbytes), dev, self.lib.bn_act_mask, raw, u.bnp, res, rres, rbnp, y, u.mask_bits, M, u.cout, mpg, 1 if relu else 0,
                   self.stream(dev))
        return y

    # ------------------------------------------------------------------ backward primitives
    def bn_bwd(self, u, g, ymask, raw, M, G, want_gm=False, relu=False):
        """gradient wrt the raw conv output (and optionally the ReLU-masked incoming gradient).
        ymask: the unit's output (residual units); relu=True without ymask: conv->BN->ReLU unit,
        the mask is recomputed from raw inside the kernels."""
        if NOMASK:
            ymask = None       # what-if only (WRONG gradients): the step without the mask operand reads
        if not u.bn.training:
            return self._bn_bwd_eval(u, g, ymask, raw, M, want_gm, relu)
        dev = raw.device
        s = self.stream(dev)
        lib = self.lib
        C = u.cout
        mpg = M // G
        ppb = math.gcd(mpg, 512)
        if ppb < 16:
            ppb = mpg
        nblk = M // ppb
        partial = self.ws('ws.bnbwd', nblk * 2 * C, torch.float32, dev)
        u.bsums = self.buf(f'{u.name}.bsums', (G, 2, C), torch.float64, dev)
        bits = ymask is not None and ymask.dtype == torch.uint8      # bit-packed mask from bn_act(want_mask=True)
        rl = 2 if bits else (1 if relu else 0)
        mask_bytes = 0.0 if ymask is None else (M * C / 8.0 if bits else 2.0 * M * C)
        fused = getattr(self, '_fused_bn', None)
        self._fused_bn = None
        raw_row = False
        if fused is not None and fused[0] is u:     # the producing dgrad already emitted the statistics rows
            partial, nblk = fused[1], fused[2]
        elif (FIN_FUSE and not self.collectives_on and nblk == G and mpg <= 512 and os.environ.get('VFS_HEAD_FUSE', '1') == '1'):
            raw_row = True      # round 6: one statistics row per group (the head's BatchNorm1d layers): the apply pass computes it itself
        else:
            self.timed('bn_bwd_reduce', (0.0, 2.0 * M * C * 2 + mask_bytes), dev, lib.bn_bwd_reduce, g, ymask, raw, u.bnp,
                       partial, M, C, mpg, ppb, rl, s)
        dx = self.buf(f'{u.name}.dx', raw.shape, BF16, dev)
        # with the bit-packed mask the masked gradient is not materialised: its consumers take (g, mask) instead (conv_bwd add_mask,
        # the downsample unit's bn_bwd)
        want_gm = want_gm and not (bits and MASK_ADD and C % 64 == 0)      # (the mask-gated add reads whole 64-channel mask words)
        gm = self.buf(f'{u.name}.gm', raw.shape, BF16, dev) if want_gm else None
        abytes = 2.0 * M * C * (3 + want_gm) + mask_bytes      # g, raw in; dx out; activation (or its bit mask) in; masked gradient out
        if raw_row:
            self.timed('bn_bwd_apply', (0.0, abytes + 2.0 * M * C * 2 + mask_bytes), dev, lib.bn_bwd_apply_raw, g, ymask, raw, u.bnp, u.bsums, u.bn.weight.grad, u.bn.bias.grad,
                       dx, gm, M, C, mpg, float(mpg), rl, s)
            return dx, gm
        if FIN_FUSE and FIN_XCHG and self.collectives_on and nblk // G <= FIN_MAX_ROWS and G * 2 * min(C, 64) <= 256 and C <= 4096:
            x = self.p2p_exchange(dev)
            if x is not None and x.fits(u.bsums):
                # SyncBN, few statistics rows: the apply pass sums them in its prologue, exchanges the sums with the peers there and
                # writes bsums (over the ranks), dgamma, dbeta (local)
                self.timed('bn_bwd_apply', (0.0, abytes), dev, lib.bn_bwd_apply_fin_xchg, g, ymask, raw, u.bnp, partial, nblk // G, u.bsums, u.bn.weight.grad, u.bn.bias.grad,
                           dx, gm, M, C, mpg, float(mpg * self.world), rl, *x.tail_args(), self.next_xseq(), s)
                return dx, gm
        if FIN_FUSE and not self.collectives_on and nblk // G <= FIN_MAX_ROWS:
            # few statistics rows: the apply pass sums them in its prologue (and writes bsums, dgamma, dbeta)
            self.timed('bn_bwd_apply', (0.0, abytes), dev, lib.bn_bwd_apply_fin, g, ymask, raw, u.bnp, partial, nblk // G, u.bsums, u.bn.weight.grad, u.bn.bias.grad, dx, gm, M, C,
                                 mpg, float(mpg), rl, s)
            return dx, gm
        self._bwd_sums(u, partial, G, nblk // G, C, dev)
        self.timed('bn_bwd_apply', (0.0, abytes), dev, lib.bn_bwd_apply, g, ymask, raw, u.bnp, u.bsums, dx, gm, M, C, mpg,
                   float(mpg * self.world), rl, s)
        return dx, gm

    def stats_tickets(self, dev):
        """uint32 tickets of the coarse statistics rows: zero once, every launch leaves them at zero"""
        t = self.bufs.get('ws.stats_tickets')
        if t is None or t.device != dev:
            t = torch.zeros(TILES_PER_TICKET[0], dtype=torch.int32, device=dev)
            self.bufs['ws.stats_tickets'] = t
            self.generation += 1
        return t

    def zero_sums(self, C, dev):
        """double[1][2][C] of zeros (never written): BatchNorm backward without the batch-statistic terms"""
        key = f'ws.zero_sums.{C}'
        t = self.bufs.get(key)
        if t is None or t.device != dev:
            t = torch.zeros(1, 2, C, dtype=torch.float64, device=dev)
            self.bufs[key] = t
            self.generation += 1
        return t

    def _bn_bwd_eval(self, u, g, ymask, raw, M, want_gm, relu):
        """BatchNorm in eval mode inside a training step (resnet.py:577-654: frozen_stages, norm_eval, partial_bn): the output is
        an affine map of the input with FIXED coefficients, so dx = g * mask * scale - the apply kernel with zero statistic
        sums - and, when gamma / beta are still trained (norm_eval), dgamma = sum g*mask*xhat, dbeta = sum g*mask with xhat
        from the running statistics.  One group, no SyncBN exchange (the parameter gradients are local sums like any other)."""
        dev = raw.device
        s = self.stream(dev)
        lib = self.lib
        C = u.cout
        bits = ymask is not None and ymask.dtype == torch.uint8
        rl = 2 if bits else (1 if relu else 0)
        fused = getattr(self, '_fused_bn', None)
        self._fused_bn = None
        if u.bn.weight.requires_grad or u.bn.bias.requires_grad:
            if fused is not None and fused[0] is u:
                partial, nblk = fused[1], fused[2]
            else:
                ppb = math.gcd(M, 512)
                if ppb < 16:
                    ppb = M
                nblk = M // ppb
                partial = self.ws('ws.bnbwd', nblk * 2 * C, torch.float32, dev)
                self.timed('bn_bwd_reduce', (0.0, 4.0 * M * C), dev, lib.bn_bwd_reduce, g, ymask, raw, u.bnp, partial, M, C, M, ppb, rl, s)
            scratch = self.buf(f'{u.name}.bsums', (1, 2, C), torch.float64, dev)
            self.timed('bn_stats', (0.0, 8.0 * nblk * C), dev, lib.bn_bwd_sums_paramgrad, partial, scratch, self.bn_scratch(1, C, dev),
                       u.bn.weight.grad, u.bn.bias.grad, 1, nblk, C, s)
        dx = self.buf(f'{u.name}.dx', raw.shape, BF16, dev)
        want_gm = want_gm and not (bits and MASK_ADD and C % 64 == 0)      # (the mask-gated add reads whole 64-channel mask words)
        gm = self.buf(f'{u.name}.gm', raw.shape, BF16, dev) if want_gm else None
        self.timed('bn_bwd_apply', (0.0, 2.0 * M * C * (3 + want_gm)), dev, lib.bn_bwd_apply, g, ymask, raw, u.bnp, self.zero_sums(C, dev), dx, gm,
                   M, C, M, 1.0, rl, s)
        return dx, gm

    def _bwd_sums(self, u, partial, G, bpg, C, dev):
        """partial (S1, S2) rows -> u.bsums (all-reduced for SyncBN) and dgamma / dbeta (local sums)"""
        s = self.stream(dev)
        if self.collectives_on:     # local sums + local dgamma / dbeta in one launch, then the sums over the ranks (SyncBN)
            x = self.p2p_exchange(dev)
            if x is not None and x.fits(u.bsums):      # ... in the same launch: the reduction's last workgroup runs the window exchange
                self.timed('bn_stats', (0.0, 8.0 * G * bpg * C), dev, self.lib.bn_bwd_sums_paramgrad_xchg, partial, u.bsums,
                           self.bn_scratch(G, C, dev), u.bn.weight.grad, u.bn.bias.grad, G, bpg, C, *x.tail_args(), s)
                return
            self.timed('bn_stats', (0.0, 8.0 * G * bpg * C), dev, self.lib.bn_bwd_sums_paramgrad, partial, u.bsums,
                       self.bn_scratch(G, C, dev), u.bn.weight.grad, u.bn.bias.grad, G, bpg, C, s)
            self.allreduce(u.bsums)
        else:
            self.timed('bn_stats', (0.0, 8.0 * G * bpg * C), dev, self.lib.bn_bwd_sums_paramgrad, partial, u.bsums,
                       self.bn_scratch(G, C, dev), u.bn.weight.grad, u.bn.bias.grad, G, bpg, C, s)

    def flush_counters(self):
        """materialise the lazily counted BatchNorm.num_batches_tracked buffers"""
        for u in self.units:
            n = getattr(u, 'nbt_pending', 0)
            if n and u.bn is not None:
                u.bn.num_batches_tracked += n
                u.nbt_pending = 0

    def stem_pool_bn_bwd(self, u, gp, yp, idx, raw, N, H, W, Hp, Wp, G, xpool=None):
        """BN backward of the stem through max-pool + ReLU (no full-resolution gradient tensor)."""
        dev = raw.device
        s = self.stream(dev)
        lib = self.lib
        C = u.cout
        if not u.bn.training:      # eval-mode BatchNorm (norm_eval): one group, no statistic terms in dx (zero sums)
            G = 1
        npg = N // G
        mpg_p = npg * Hp * Wp
        ppb = math.gcd(mpg_p, 256)
        if ppb < 16:
            ppb = mpg_p
        nblk = (N * Hp * Wp) // ppb
        partial = self.ws('ws.bnbwd', nblk * 2 * C, torch.float32, dev)
        u.bsums = self.buf(f'{u.name}.bsums', (G, 2, C), torch.float64, dev)
        self.timed('bn_bwd_reduce', (0.0, (2.0 * 2 + 1.0) * N * Hp * Wp * C), dev, lib.stem_pool_bn_bwd_reduce, gp, yp, idx, raw, xpool,
                   u.bnp, partial, N, H, W, C, Hp, Wp, npg, ppb, s)
        if u.bn.training:
            self._bwd_sums(u, partial, G, nblk // G, C, dev)
            return float(npg * H * W * self.world)
        if u.bn.weight.requires_grad or u.bn.bias.requires_grad:
            scratch = self.buf(f'{u.name}.bsums_eval', (1, 2, C), torch.float64, dev)
            self.timed('bn_stats', (0.0, 8.0 * nblk * C), dev, lib.bn_bwd_sums_paramgrad, partial, scratch, self.bn_scratch(1, C, dev),
                       u.bn.weight.grad, u.bn.bias.grad, 1, nblk, C, s)
        u.bsums = self.zero_sums(C, dev)
        return 1.0

    # ------------------------------------------------------------------ side stream for weight gradients
    def on_side_stream(self, dev):
        """context: subsequent launches go to the side stream, ordered after everything already
        enqueued on the current stream (no-op on the CPU / when VFS_SIDE_STREAM=0)"""
        import contextlib
        if dev.type != 'cuda' or os.environ.get('VFS_SIDE_STREAM', '1') != '1':
            return contextlib.nullcontext()
        side = self._side.get(dev)
        if side is None:
            side = self._side[dev] = torch.cuda.Stream(dev)
        self.record(side.wait_stream, torch.cuda.current_stream(dev))
        self._side_dirty = True
        return torch.cuda.stream(side)

    def wgrad_join(self, dev):
        """the current stream waits for every weight-gradient kernel issued so far"""
        if dev.type == 'cuda' and getattr(self, '_side_dirty', False):
            side = self._side.get(dev)
            if side is not None:
                self.record(torch.cuda.current_stream(dev).wait_stream, side)
            self._side_dirty = False

    def stem_wgrad_fused(self, u, x4, Hin, Win, gp, yp, idx, raw, N, H, W, Hp, Wp, G, count):
        """stem weight gradient; the BN-backward apply pass is folded into its operand load"""
        dev = raw.device
        ntiles = N * ((H + 7) // 8) * ((W + 15) // 16)
        nb = int(os.environ.get('VFS_STEM_WGRAD_BLOCKS', '1536'))
        tpb = (ntiles + nb - 1) // nb        # ~6 workgroups per CU: the operand gather is latency-bound
        nblocks = (ntiles + tpb - 1) // tpb
        partial = self.wgrad_partial(u, nblocks, 64, 224, dev)
        with self.on_side_stream(dev):      # ws.wgrad belongs to the side stream
            self.timed('stem_wgrad', (2.0 * N * H * W * 64 * 147, 2.0 * N * (Hin * Win * 4 + H * W * 64) + 5.0 * N * Hp * Wp * 64),
                       dev, self.lib.stem_wgrad_fused,
                       x4, raw, gp, yp, idx, u.bnp, u.bsums, partial, self.wgrad_target(u, partial, nblocks, 64, 224, 3, 7, 1),
                       N, Hin, Win, H, W, Hp, Wp, (N // G) if u.bn.training else N, count, nblocks, self.stream(dev))

    def wgrad_partial(self, u, nsplit, cout, ktot, dev):
        """split-K workspace of unit u's weight gradient: the shared scratch (reduced right after the kernel) or, when the
        reduction is deferred, a per-unit buffer that lives until flush_wgrad"""
        if self.defer_wgrad:
            return self.buf(f'{u.name}.wpart', (nsplit * cout * ktot,), torch.float32, dev)
        return self.ws('ws.wgrad', nsplit * cout * ktot, torch.float32, dev)

    def wgrad_tickets(self, dev):
        """uint32 tickets of the in-launch split-K reductions (vfs_conv_wgrad_inl): zero once, every launch leaves them at zero;
        shared by all weight-gradient launches - they run one after the other on the side stream"""
        t = self.bufs.get('ws.wgrad_tickets')
        if t is None or t.device != dev:
            t = torch.zeros(self.n_wgrad_tickets, dtype=torch.int32, device=dev)
            self.bufs['ws.wgrad_tickets'] = t
            self.generation += 1
        return t

    def wgrad_target(self, u, partial, nsplit, cout, ktot, cin, k, stem):
        """the `grad` argument of the weight-gradient entry points: the gradient itself, or None (= reduce later)"""
        if not self.defer_wgrad:
            return u.weight.grad
        self._wpending.append((partial, u.weight.grad, nsplit, cout, ktot, cin, k, k, stem))
        return None

    def flush_wgrad(self, dev):
        """ONE table-driven launch reduces the split-K partials of every weight gradient issued since the last flush
        (side stream, after the kernels that wrote them)"""
        if not self._wpending:
            return
        pend, self._wpending = self._wpending, []
        key = tuple((p.data_ptr(), g.data_ptr(), ns) for p, g, ns, *_ in pend)
        tab = self._wtables.get(key)
        if tab is None:
            if len(self._wtables) > 64:
                self._wtables.clear()
            tab = self._wtables[key] = build_reduce_table(pend, dev)
            self.generation += 1
        with self.on_side_stream(dev):
            self.lib.wgrad_reduce_table(tab[0], tab[1], tab[2], self.stream(dev))

    def conv_bwd(self, u, dx, x_in, N, H, W, Ho, Wo, need_dgrad, add=None, g_out=None, bn_next=None, x_in_bn=None, add_mask=None):
        """weight (and bias) gradients accumulate into .grad; returns the input gradient or None.
        bn_next = (unit, raw, ymask, relu, G): the BatchNorm unit whose backward consumes the input
        gradient; for stride-1 convs the dgrad epilogue also emits that unit's backward statistics
        (vfs_conv_dgrad_bn) and the following bn_bwd skips its reduce pass."""
        dev = dx.device
        s = self.stream(dev)
        lib = self.lib
        M = N * Ho * Wo
        if u.dil != 1:
            raise NotImplementedError(f'{u.name}: dilated convolutions are forward-only here (the SiamFC probe freezes its '
                                      'dilated backbone: frozen_stages=4, norm_eval=True)')
        if u.kind == 'stem':
            nsplit, pps = wgrad_splits(M, 64, 256)
            partial = self.wgrad_partial(u, nsplit, 64, 256, dev)
            with self.on_side_stream(dev):
                self.timed('stem_wgrad', (2.0 * M * 64 * 147, 2.0 * (M * 64 + N * H * W * 4)), dev, lib.stem_wgrad,
                           dx, x_in, partial, self.wgrad_target(u, partial, nsplit, 64, 256, 3, 7, 1), N, H, W, Ho, Wo, nsplit, pps,
                           self.stream(dev))
            return None
        ktot = u.k * u.k * u.cin
        halo = (N, H, W, u.cin) if wgrad_halo_eligible(N, H, W, u.cin, u.cout, u.k, u.stride, u.pad) else None
        nsplit, pps = wgrad_splits(M, u.cout, ktot, halo_geom=halo)
        inl = (WGRAD_INL and not self.defer_wgrad and u.cin % 4 == 0 and ((ktot + 127) // 128) * (u.cout // 64) <= self.n_wgrad_tickets
               and not (x_in_bn is not None and halo is None))      # (the in-launch reduction folds the input BatchNorm in the halo kernel only)
        partial = ((self.ws('ws.wgrad', wgrad_inl_floats(nsplit, u.cout, ktot), torch.float32, dev) if inl else self.wgrad_partial(u, nsplit, u.cout, ktot, dev))
                   if u.weight.requires_grad else None)
        flops = 2.0 * M * u.cout * ktot
        # ALGORITHMIC bytes: dY + x once, the fp32 gradient read-modify-write.  (The fp32 split-K partials - written by the
        # kernel, re-read by the reduction: 8 * nsplit * Cout * Ktot bytes - are implementation traffic; they show up in the
        # PMC 'traffic' figure, not here.)
        wbytes = 2.0 * (M * u.cout + N * H * W * u.cin) + 8.0 * u.cout * ktot
        # dgrad: dY + weights in, dx out (+ residual gradient / fused BatchNorm operands when present)
        dbytes = 2.0 * (M * u.cout + N * H * W * u.cin + u.cout * ktot)
        # the weight gradient only feeds the optimizer: it runs on the side stream, concurrently
        # with the dgrad / BatchNorm-backward kernels of the critical path (joined by wgrad_join).  Frozen weights
        # (requires_grad = False: frozen_stages) get none.
        if u.weight.requires_grad:
            wtarget = self.wgrad_target(u, partial, nsplit, u.cout, ktot, u.cin, u.k, 0)
            with self.on_side_stream(dev):
                ss = self.stream(dev)
                if inl:
                    # one launch: the last workgroup of a tile sums the split-K partials itself (the launches of ONE stream share the tickets)
                    inb = x_in_bn if x_in_bn is not None else (None, 0)
                    self.timed('conv3x3_wgrad_halo' if halo is not None else 'conv_wgrad', (flops, wbytes), dev, lib.conv_wgrad_inl, dx, x_in, inb[0], inb[1],
                               partial, wtarget, self.wgrad_tickets(dev), N, H, W, u.cin, Ho, Wo, u.cout, u.k, u.k, u.stride, u.pad, nsplit, pps, ss)
                elif x_in_bn is not None:     # x_in is the producer's RAW output (see conv_fwd)
                    self.timed('conv3x3_wgrad_halo' if halo is not None else 'conv_wgrad', (flops, wbytes), dev, lib.conv_wgrad_bnin, dx, x_in, x_in_bn[0], x_in_bn[1], partial,
                               wtarget, N, H, W, u.cin, Ho, Wo, u.cout, u.k, u.k, u.stride, u.pad, nsplit, pps, ss)
                else:
                    self.timed('conv3x3_wgrad_halo' if halo is not None else 'conv_wgrad', (flops, wbytes), dev, lib.conv_wgrad, dx, x_in, partial, wtarget, N, H, W, u.cin, Ho,
                               Wo, u.cout, u.k, u.k, u.stride, u.pad, nsplit, pps, ss)
                if u.bias is not None and u.bias.requires_grad:
                    lib.bias_grad(dx, u.bias.grad, M, u.cout, ss)
        if not need_dgrad:
            return None
        gin = g_out if g_out is not None else self.buf(f'{u.name}.gin', (N, H, W, u.cin), BF16, dev)
        self._fused_bn = None
        ks, ksws = igemm_ksplit(N * H * W, u.cin, ktot // u.cin * u.cout) if (u.k == 1 and u.stride == 1 and KSPLIT) else (1, 0)
        if ks == 1 and bn_next is not None and u.stride == 1 and os.environ.get('VFS_BN_FUSE', '1') == '1':
            pu, praw, pymask, prelu, G = bn_next
            if NOMASK:
                pymask = None
            Min = N * H * W
            mpg = Min // G
            # the dgrad as a conv [N,Ho,Wo,cout] -> [N,H,W,cin]: its statistics rows (tiles or linear blocks)
            rows = conv_stats_rows(N, G, Ho, Wo, u.cout, u.cin, u.k, u.stride, u.pad, H, W)
            if rows is not None:
                nblk = rows * G
                partial = self.ws('ws.bnbwd_fused', nblk * 2 * u.cin, torch.float32, dev)
                pbits = pymask is not None and pymask.dtype == torch.uint8
                mask_units = 0.0 if pymask is None else (1.0 / 16 if pbits else 1.0)
                self.timed(self.conv_kind(u, N, Ho, Wo, dgrad=True), (flops, dbytes + 2.0 * N * H * W * u.cin * ((1 if add is not None else 0) + 1 + mask_units)),
                           dev, lib.conv_dgrad_bn_maskadd, dx, u.wd, gin, add, add_mask, praw, pymask, pu.bnp, partial,
                           mpg, 2 if pbits else (1 if (prelu and pymask is None) else 0), N, H, W, u.cin, Ho, Wo, u.cout, u.k, u.k,
                           u.stride, u.pad, s)
                self._fused_bn = (pu, partial, nblk)
                return gin
        work = (flops, dbytes + (2.0 * N * H * W * u.cin if add is not None else 0.0))
        if ks > 1:
            assert add_mask is None, 'the split-K dgrad has no mask-gated add'
            self.timed('conv_igemm', work, dev, lib.conv_dgrad_splitk, dx, u.wd, gin, add, self.ksplit_ws(ksws, dev), ks, N, H, W, u.cin,
                       Ho, Wo, u.cout, u.k, u.k, u.stride, u.pad, s)
        else:
            self.timed(self.conv_kind(u, N, Ho, Wo, dgrad=True), work, dev, lib.conv_dgrad_maskadd, dx, u.wd, gin, add, add_mask, N, H, W, u.cin, Ho, Wo, u.cout,
                       u.k, u.k, u.stride, u.pad, s)
        return gin


_ENGINES = {}

# Bumped by everything that writes parameters or BatchNorm running statistics THROUGH RAW POINTERS (vfs_sgd_step /
# vfs_adam_step on the arenas, the BatchNorm kernels of a training forward): such writes never touch tensor._version, so
# caches of derived data (the fp32 evaluation executor's folded BatchNorm / repacked weights, the SiamFC head's packed
# weights) key on this counter as well.
_PARAMS_EPOCH = [0]


def params_epoch():
    return _PARAMS_EPOCH[0]


def bump_params_epoch():
    _PARAMS_EPOCH[0] += 1


def shared_engine(device=None):
    """One Engine (buffer pool + packed weights) per process; one process drives one GPU."""
    eng = _ENGINES.get('default')
    if eng is None:
        eng = _ENGINES['default'] = Engine()
    return eng


def flush_counters_hook(module, prefix, keep_vars):
    """state_dict pre-hook of the VFS modules: num_batches_tracked is counted on the host"""
    eng = _ENGINES.get('default')
    if eng is not None:
        eng.flush_counters()


def set_shared_engine(eng):
    _ENGINES['default'] = eng
