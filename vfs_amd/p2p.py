"""SyncBN statistic exchange over xGMI (csrc/p2p.hip): the host side - window allocation, IPC handshake, self-test.

The reference's SyncBN (configs/r*_*.py:9,15; MMDistributedDataParallel, apis/train.py:58-66) all-reduces a few KB per BatchNorm
layer and direction: 114 dependent, latency-bound collectives per ResNet-50 step.  `P2PExchange` replaces each of them by one small
kernel (`vfs_p2p_allreduce_f64`): every rank stores its numbers into IPC-mapped windows of all ranks, stamps a flag, waits for the
peers' stamps in its own window and adds the contributions in rank order.  torch.distributed is only used ONCE, to exchange the
64-byte IPC handles (and to agree on the outcome of the self-test).  The bandwidth-bound gradient buckets stay on RCCL.

One node only (<= 8 ranks, <= 8192 doubles per exchange); anything else, and any failure of the set-up or the self-test on ANY
rank, leaves the engine on the collective-library path."""
import os

import torch
import torch.distributed as dist


class P2PExchange:
    def __init__(self, lib, dev, group=None, spin_limit=None):
        self.lib, self.dev, self.group = lib, dev, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # polls (each followed by s_sleep 8, ~0.3 us) before a waiting kernel gives up: ~10 minutes by default - a rank whose host
        # stalls (checkpoint, data loader) must not turn into silently wrong statistics; tests use a short limit
        self.spin_limit = int(spin_limit if spin_limit is not None else os.environ.get('VFS_P2P_SPIN', str(1 << 31)))
        meta = torch.zeros(3, dtype=torch.int64)
        i32 = torch.zeros(2, dtype=torch.int32)
        lib.p2p_window_bytes(meta, i32[0:1], i32[1:2])
        self.window_bytes, self.max_doubles, self.max_world = int(meta[0]), int(i32[0]), int(i32[1])
        if self.world > self.max_world:
            raise RuntimeError(f'P2PExchange: {self.world} ranks > {self.max_world} (one node)')
        # Local part first (allocation, export) - it can fail on ONE rank only (out of fine-grained memory, IPC refused).  Every rank
        # then takes part in the SAME all_gather whatever happened locally, and peers' windows are only mapped when every rank
        # reported success: a rank that raised before the collective used to leave the others blocked in it.
        self.window, self._imported = 0, []
        out = torch.zeros(1, dtype=torch.int64)
        handle, err = None, None
        try:
            lib.p2p_alloc(out)
            self.window = int(out[0])
            hb = torch.zeros(64, dtype=torch.uint8)
            lib.p2p_export(self.window, hb)
            handle = bytes(hb.numpy().tobytes())
        except Exception as e:      # noqa: BLE001 - reported to the peers below, raised after the collective
            err = e
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (self.rank, os.getpid(), handle), group=group)
        try:
            if err is not None:
                raise err
            bad = [r for r, _, hb in gathered if hb is None]
            if bad:
                raise RuntimeError(f'P2PExchange: window set-up failed on rank(s) {bad}')
            ptrs = []
            for r, pid, hb in gathered:
                if r == self.rank:
                    ptrs.append(self.window)
                    continue
                h = torch.frombuffer(bytearray(hb), dtype=torch.uint8)
                lib.p2p_import(h, out)
                ptrs.append(int(out[0]))
                self._imported.append(int(out[0]))
        except Exception:
            self.close()            # the window and whatever was mapped so far
            raise
        self.peers = torch.tensor(ptrs, dtype=torch.int64, device=dev)
        self.state = torch.zeros(4 + 64, dtype=torch.int64, device=dev)      # {exchange counter, error flag, workgroup ticket, -, slab_ready[64] of the folded exchanges (vfs_p2p.h)}

    def allreduce(self, lib, t, stream):
        """in place: t <- sum over ranks (rank order; bit-identical everywhere).  `lib` = the engine's current library object,
        so that a recording command tape sees the call"""
        lib.p2p_allreduce_f64(t, t.numel(), self.peers, self.rank, self.world, self.state, 3, self.spin_limit, stream)

    def tail_args(self):
        """the trailing arguments of the vfs_*_xchg entry points (reductions that run the exchange as their tail)"""
        return (self.peers, self.rank, self.world, self.state, self.spin_limit)

    def fits(self, t):
        return t.dtype == torch.float64 and t.is_contiguous() and 0 < t.numel() <= self.max_doubles and t.device == self.dev

    def failed(self):
        return bool(int(self.state[1].item()))

    def raise_if_failed(self):
        if self.failed():
            raise RuntimeError('vfs_p2p_allreduce_f64: a peer did not arrive within the spin limit (VFS_P2P_SPIN); the statistics of '
                               'this step are invalid')

    def self_test(self, stream=None, rounds=3):
        """a few exchanges of known vectors with a short spin limit; True only if EVERY rank saw the right sums"""
        keep, self.spin_limit = self.spin_limit, 1 << 22
        ok = True
        try:
            for k in range(rounds):
                n = (1, 257, self.max_doubles)[k % 3]
                t = (torch.arange(n, dtype=torch.float64, device=self.dev) + 1.0) * (self.rank + 1 + k)
                self.allreduce(self.lib, t, stream)
                want = (torch.arange(n, dtype=torch.float64, device=self.dev) + 1.0) * sum(r + 1 + k for r in range(self.world))
                ok = ok and bool(torch.equal(t, want))
            ok = ok and not self.failed()
        except Exception:      # noqa: BLE001 - any failure here means "use the collective library"
            ok = False
        finally:
            self.spin_limit = keep
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.dev if dist.get_backend(self.group) == 'nccl' else 'cpu')
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def close(self):
        for p in self._imported:
            self.lib.p2p_unimport(p)
        self._imported = []
        if self.window:
            self.lib.p2p_free(self.window)
            self.window = 0
