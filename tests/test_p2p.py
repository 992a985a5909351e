"""SyncBN statistic exchange through IPC-mapped windows (csrc/p2p.hip, vfs_amd/p2p.py).
  * emulator: the protocol with two "ranks" = two host threads of this process (the emulator's IPC handle is the pointer);
  * GPU: TWO PROCESSES sharing the one GPU of the box (hipIpcGetMemHandle / hipIpcOpenMemHandle between real processes - RCCL
    refuses two ranks on one device, IPC does not): protocol soak vs a gloo all-reduce, and data-parallel train steps whose
    SyncBN statistics travel through the windows vs the same steps with collective-library all-reduces - bit-equal."""
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _windows(lib, world):
    out = torch.zeros(1, dtype=torch.int64)
    wins = []
    for _ in range(world):
        lib.p2p_alloc(out)
        wins.append(int(out[0]))
    # export / import round trip (in the emulator the handle carries the pointer)
    mapped = []
    for w in wins:
        h = torch.zeros(64, dtype=torch.uint8)
        lib.p2p_export(w, h)
        lib.p2p_import(h, out)
        mapped.append(int(out[0]))
    return wins, mapped


def test_protocol_two_threads_emulator(emu_backend):
    lib = emu_backend.lib
    meta, i32 = torch.zeros(1, dtype=torch.int64), torch.zeros(2, dtype=torch.int32)
    lib.p2p_window_bytes(meta, i32[0:1], i32[1:2])
    assert int(i32[0]) == 8192 and int(i32[1]) == 8 and int(meta[0]) > 8192 * 8 * 8
    world = 3
    wins, mapped = _windows(lib, world)
    peers = torch.tensor(mapped, dtype=torch.int64)
    sizes = [1, 7, 256, 1000, 8192, 33, 4096, 2, 513, 64]      # more exchanges than slots: the ring wraps
    g = torch.Generator().manual_seed(0)
    data = [[torch.randn(n, generator=g, dtype=torch.float64) for n in sizes] for _ in range(world)]
    results = [[None] * len(sizes) for _ in range(world)]
    states = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]

    def rank_main(r):
        for k, n in enumerate(sizes):
            t = data[r][k].clone()
            lib.p2p_allreduce_f64(t, n, peers, r, world, states[r], 3, 1 << 40, None)
            results[r][k] = t
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive()
    for k in range(len(sizes)):
        want = data[0][k].clone()
        for r in range(1, world):
            want = want + data[r][k]          # rank order
        for r in range(world):
            assert torch.equal(results[r][k], want), (r, k)
    for st in states:
        assert int(st[0]) == len(sizes) and int(st[1]) == 0
    # a lost peer: bounded spin, error word set, no hang
    lone = torch.ones(4, dtype=torch.float64)
    lib.p2p_allreduce_f64(lone, 4, peers, 0, world, states[0], 3, 1000, None)
    assert int(states[0][1]) == 1
    assert torch.isnan(lone).all()      # the sums of a failed exchange are poisoned, not left as plausible garbage
    for w in wins:
        lib.p2p_free(w)


def test_protocol_eight_threads_emulator(emu_backend):
    """the largest configuration the windows hold: 8 ranks (P2P_MAXW) x 8192 doubles (P2P_MAXN), more exchanges than ring slots,
    every rank's result bit-equal to the rank-ordered sum"""
    lib = emu_backend.lib
    world = 8
    wins, mapped = _windows(lib, world)
    peers = torch.tensor(mapped, dtype=torch.int64)
    sizes = [8192, 1, 8192, 4097, 8192, 63, 8192]
    g = torch.Generator().manual_seed(8)
    data = [[torch.randn(n, generator=g, dtype=torch.float64) for n in sizes] for _ in range(world)]
    results = [[None] * len(sizes) for _ in range(world)]
    states = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]

    def rank_main(r):
        for k, n in enumerate(sizes):
            t = data[r][k].clone()
            lib.p2p_allreduce_f64(t, n, peers, r, world, states[r], 3, 1 << 40, None)
            results[r][k] = t
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    for k in range(len(sizes)):
        want = data[0][k].clone()
        for r in range(1, world):
            want = want + data[r][k]
        for r in range(world):
            assert torch.equal(results[r][k], want), (r, k)
    for st in states:
        assert int(st[0]) == len(sizes) and int(st[1]) == 0
    for w in wins:
        lib.p2p_free(w)


@pytest.mark.parametrize('G,bpg,C', [(2, 8, 64), (2, 40, 256), (2, 300, 128), (1, 130, 2048)])
def test_reductions_with_exchange_tail_two_threads_emulator(emu_backend, G, bpg, C):
    """vfs_bn_reduce_partials_xchg / vfs_bn_bwd_sums_paramgrad_xchg (the last workgroup of the reduction runs the exchange): two
    "ranks" on two host threads, each with its own statistics rows - every rank ends with the SUM of both ranks' plain
    reductions (rank order), local dgamma / dbeta untouched by the exchange; small (one workgroup per channel block), medium and
    ticketed (> 64 rows per group) reductions, C up to 2048 (64 channel blocks: every ticket counter in use)"""
    lib = emu_backend.lib
    world = 2
    wins, mapped = _windows(lib, world)
    peers = torch.tensor(mapped, dtype=torch.int64)
    g = torch.Generator().manual_seed(G * 1000 + bpg)
    rows = [torch.randn(G * bpg, 2, C, generator=g) for _ in range(world)]
    plain = []
    for r in range(world):
        sums = torch.zeros(G, 2, C, dtype=torch.float64)
        lib.bn_reduce_partials(rows[r], sums, torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64), G, bpg, C, None)
        plain.append(sums)
    want = plain[0] + plain[1]
    for mode in ('fwd', 'bwd'):
        states = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
        out = [torch.zeros(G, 2, C, dtype=torch.float64) for _ in range(world)]
        dg = [torch.zeros(C) for _ in range(world)]
        db = [torch.zeros(C) for _ in range(world)]

        def rank_main(r):
            scratch = torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64)
            for _ in range(3):      # repeated launches: tickets and counters must return to zero
                if mode == 'fwd':
                    lib.bn_reduce_partials_xchg(rows[r], out[r], scratch, G, bpg, C, peers, r, world, states[r], 1 << 40, None)
                else:
                    dg[r].zero_(), db[r].zero_()
                    lib.bn_bwd_sums_paramgrad_xchg(rows[r], out[r], scratch, dg[r], db[r], G, bpg, C, peers, r, world, states[r], 1 << 40, None)
        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
            assert not t.is_alive()
        for r in range(world):
            assert torch.equal(out[r], want), (mode, r)
            assert int(states[r][0]) == 3 and int(states[r][1]) == 0 and int(states[r][2]) == 0
            if mode == 'bwd':      # dbeta = sum over groups of S1, dgamma of S2 - LOCAL sums (they travel with the gradient buckets)
                assert torch.allclose(db[r].double(), plain[r][:, 0].sum(0), rtol=1e-6, atol=1e-6)
                assert torch.allclose(dg[r].double(), plain[r][:, 1].sum(0), rtol=1e-6, atol=1e-6)
    for w in wins:
        lib.p2p_free(w)


@pytest.mark.parametrize('world,G,rows,C,ppr', [(2, 2, 6, 128, 32), (3, 2, 16, 256, 128), (2, 1, 33, 64, 16), (2, 2, 3, 32, 64)])
def test_apply_passes_with_folded_exchange_threads_emulator(emu_backend, world, G, rows, C, ppr):
    """round 6: vfs_bn_act_fin_xchg / vfs_bn_bwd_apply_fin_xchg - the window exchange folded into the apply pass (the first
    workgroup of every 64-channel slab exchanges its slab's sums with the peers' workgroups of the same slab, the others wait on a
    device-scope word).  `world` ranks on host threads, each with its own tensor and statistics rows.  Expected results from the
    unfolded pieces: the rank's LOCAL sums (vfs_bn_act_fin on its own rows), added over the ranks in rank order, then
    vfs_bn_act_fin(partial = NULL) / vfs_bn_bwd_apply on those totals with the global count - the folded launch must give the
    same bits (outputs, bnp, sums, running statistics; backward: dx, gm, sums; dgamma / dbeta stay LOCAL sums).  Three launch
    pairs with rising sequence numbers (the host numbers the folded exchanges of a chain), then the same numbers again behind a
    vfs_p2p_chain_start - what a replayed launch chain does."""
    from tests.emu_util import rb
    lib = emu_backend.lib
    wins, mapped = _windows(lib, world)
    peers = torch.tensor(mapped, dtype=torch.int64)
    g = torch.Generator().manual_seed(world * 100 + rows * 7 + C)
    mpg = rows * ppr
    M = G * mpg
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    xs, parts, ress, gys, bparts = [], [], [], [], []
    for r in range(world):
        x = rb(torch.randn(M, C, generator=g) * 1.3 + 0.2 * r)
        xs.append(x.to(torch.bfloat16))
        parts.append(torch.stack([x.view(G * rows, ppr, C).sum(1), (x * x).view(G * rows, ppr, C).sum(1)], dim=1).contiguous())
        ress.append(rb(torch.randn(M, C, generator=g)).to(torch.bfloat16))
        gys.append(rb(torch.randn(M, C, generator=g)).to(torch.bfloat16))
        bparts.append((torch.randn(G * rows, 2, C, generator=g) * 3).contiguous())
    cnt = float(mpg * world)
    # ---- expected, from the unfolded pieces
    loc, bloc = [], []
    for r in range(world):
        sums, bs = torch.zeros(G, 2, C, dtype=torch.float64), torch.zeros(G, 2, C, dtype=torch.float64)
        lib.bn_act_fin(xs[r], parts[r], rows, gamma, beta, torch.zeros(G, 4, C), sums, torch.zeros(C), torch.ones(C), None, None, None,
                       torch.empty(M, C, dtype=torch.bfloat16), M, C, mpg, 1, float(mpg), 1e-5, 0.1, None)
        lib.bn_bwd_apply_fin(gys[r], None, xs[r], torch.ones(G, 4, C), bparts[r], rows, bs, torch.zeros(C), torch.zeros(C),
                             torch.empty(M, C, dtype=torch.bfloat16), None, M, C, mpg, float(mpg), 1, None)
        loc.append(sums)
        bloc.append(bs)
    tot, btot = loc[0].clone(), bloc[0].clone()
    for r in range(1, world):
        tot, btot = tot + loc[r], btot + bloc[r]
    want = []
    for r in range(world):
        rm, rv, bnp = torch.full((C,), 0.25), torch.full((C,), 1.5), torch.zeros(G, 4, C)
        y = torch.empty(M, C, dtype=torch.bfloat16)
        lib.bn_act_fin(xs[r], None, 0, gamma, beta, bnp, tot.clone(), rm, rv, ress[r], None, None, y, M, C, mpg, 1, cnt, 1e-5, 0.1, None)
        dx, gm = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
        lib.bn_bwd_apply(gys[r], None, xs[r], bnp, btot, dx, gm, M, C, mpg, cnt, 1, None)
        want.append((y, bnp, rm, rv, dx, gm))
    # ---- the folded launches, one thread per rank
    states = [torch.zeros(4 + 64, dtype=torch.int64) for _ in range(world)]
    got = [None] * world

    def rank_main(r):
        for it in range(6):
            if it == 3:      # a second "chain" with the same sequence numbers, as a replayed launch chain issues them
                lib.p2p_chain_start(states[r], None)
            rm, rv, bnp, sums = torch.full((C,), 0.25), torch.full((C,), 1.5), torch.zeros(G, 4, C), torch.zeros(G, 2, C, dtype=torch.float64)
            y = torch.empty(M, C, dtype=torch.bfloat16)
            lib.bn_act_fin_xchg(xs[r], parts[r], rows, gamma, beta, bnp, sums, rm, rv, ress[r], None, None, y, None, M, C, mpg, 1, cnt, 1e-5, 0.1,
                                peers, r, world, states[r], 1 << 40, 2 * (it % 3), None)
            bs, dg, db = torch.zeros(G, 2, C, dtype=torch.float64), torch.full((C,), 0.5), torch.full((C,), -0.25)
            dx, gm = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
            lib.bn_bwd_apply_fin_xchg(gys[r], None, xs[r], bnp, bparts[r], rows, bs, dg, db, dx, gm, M, C, mpg, cnt, 1,
                                      peers, r, world, states[r], 1 << 40, 2 * (it % 3) + 1, None)
            got[r] = (y, bnp, rm, rv, dx, gm, sums, bs, dg, db)
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    for r in range(world):
        y, bnp, rm, rv, dx, gm, sums, bs, dg, db = got[r]
        wy, wbnp, wrm, wrv, wdx, wgm = want[r]
        assert torch.equal(sums, tot) and torch.equal(bs, btot), r
        assert torch.equal(bnp, wbnp) and torch.equal(rm, wrm) and torch.equal(rv, wrv), r
        assert torch.equal(y.view(torch.int16), wy.view(torch.int16)), r
        assert torch.equal(dx.view(torch.int16), wdx.view(torch.int16)) and torch.equal(gm.view(torch.int16), wgm.view(torch.int16)), r
        assert torch.allclose(db.double(), -0.25 + bloc[r][:, 0].sum(0), rtol=1e-6, atol=1e-6)      # local sums
        assert torch.allclose(dg.double(), 0.5 + bloc[r][:, 1].sum(0), rtol=1e-6, atol=1e-6)
        assert int(states[r][0]) == 0 and int(states[r][1]) == 0 and int(states[r][3]) == 1      # (the folded exchanges have their own epoch sequence: chain counter, host-assigned numbers)
    # a lost peer: rank 0 alone, bounded spin -> error word, NaN statistics, no hang
    lone_state = torch.zeros(4 + 64, dtype=torch.int64)
    bnp, sums = torch.zeros(G, 4, C), torch.zeros(G, 2, C, dtype=torch.float64)
    lib.bn_act_fin_xchg(xs[0], parts[0], rows, gamma, beta, bnp, sums, torch.zeros(C), torch.ones(C), None, None, None,
                        torch.empty(M, C, dtype=torch.bfloat16), None, M, C, mpg, 1, cnt, 1e-5, 0.1, peers, 0, world, lone_state, 200, 9, None)
    assert int(lone_state[1]) == 1 and torch.isnan(sums).all()
    for w in wins:
        lib.p2p_free(w)


def test_argument_checks(emu_backend):
    from vfs_amd._lib import VfsError
    lib = emu_backend.lib
    wins, mapped = _windows(lib, 1)
    peers, st = torch.tensor(mapped, dtype=torch.int64), torch.zeros(2, dtype=torch.int64)
    t = torch.zeros(8193, dtype=torch.float64)
    with pytest.raises(VfsError):
        lib.p2p_allreduce_f64(t, 8193, peers, 0, 1, st, 3, 10, None)
    with pytest.raises(VfsError):
        lib.p2p_allreduce_f64(t, 4, peers, 1, 1, st, 3, 10, None)
    with pytest.raises(VfsError):
        lib.p2p_allreduce_f64(t, 4, peers, 0, 9, st, 3, 10, None)
    lib.p2p_allreduce_f64(t, 4, peers, 0, 1, st, 3, 10, None)      # world 1: the identity
    assert int(st[0]) == 1 and int(st[1]) == 0
    lib.p2p_free(wins[0])


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_two(tmp_path, tag, mode, extra_env):
    port = str(_free_port())
    procs, outs = [], []
    for r in range(2):
        o = str(tmp_path / f'{tag}_rank{r}.npz')
        outs.append(o)
        env = dict(os.environ, WORLD_SIZE='2', RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY='0',
                   VFS_P2P_SPIN=str(1 << 24), **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, 'tests', 'p2p_worker.py'), mode, o], env=env))
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert rcs == [0, 0], rcs
    return [np.load(o) for o in outs]


@pytest.mark.gpu
def test_two_processes_one_gpu_protocol(gpu_backend, tmp_path):
    """300 exchanges of random sizes between two processes on cuda:0, every result compared with the sum computed from both
    ranks' inputs (gathered through gloo): bit-equal on both ranks"""
    r = _run_two(tmp_path, 'proto', 'protocol', {})
    assert int(r[0]['exchanges']) == int(r[1]['exchanges']) == 300
    assert int(r[0]['mismatches']) == 0 and int(r[1]['mismatches']) == 0
    assert int(r[0]['error_word']) == 0 and int(r[1]['error_word']) == 0
    print('two processes, one GPU: us per exchange (kernel launch to completion, host-timed, 4 KB payload):', float(r[0]['us_per_exchange']))


@pytest.mark.gpu
def test_two_processes_one_gpu_train_steps_equal_collective_path(gpu_backend, tmp_path):
    """3 data-parallel steps (eager, recorded, replayed) of the shallow R18 on two processes sharing cuda:0: SyncBN statistics through
    the P2P windows vs through gloo all-reduces (gradients through gloo in both) - every parameter, gradient and running statistic
    bit-equal, and the two ranks in lock-step"""
    p2p = _run_two(tmp_path, 'p2p', 'train', dict(VFS_SYNCBN_P2P='1', VFS_FIN_XCHG='0', VFS_TEST_STEPS='3'))      # (the exchange as the tail of the reduction launches: the same summation order as the collective path)
    coll = _run_two(tmp_path, 'coll', 'train', dict(VFS_SYNCBN_P2P='0', VFS_TEST_STEPS='3'))
    assert int(p2p[0]['p2p_active']) == 1 and int(coll[0]['p2p_active']) == 0
    assert int(p2p[0]['p2p_exchanges']) > 0
    n = 0
    for k in coll[0].files:
        if k.startswith(('param/', 'grad/', 'buf/', 'log/')):
            assert np.array_equal(p2p[0][k], coll[0][k]), k
            n += 1
        if k.startswith(('param/', 'grad/', 'buf/')):
            assert np.array_equal(p2p[0][k], p2p[1][k]), k
    assert n > 50


@pytest.mark.gpu
def test_two_processes_one_gpu_train_steps_folded_exchange(gpu_backend, tmp_path):
    """round 6, the default N > 1 path: the window exchange folded into the BatchNorm apply launches (vfs_bn_act_fin_xchg /
    vfs_bn_bwd_apply_fin_xchg - the slab leads of the two processes exchange their sums, the other workgroups wait on a device-scope
    word).  3 data-parallel steps (eager, recorded, replayed): the two ranks stay in lock-step BIT FOR BIT (every rank adds the same
    numbers in the same order), and the run equals the collective-library run up to the summation order of the statistics rows
    (the apply pass's prologue sums them four rows at a time, the reduction launch eight: fp64 sums of fp32 rows, last-bit
    differences) - parameters to 1e-5 of their scale after three steps."""
    fold = _run_two(tmp_path, 'fold', 'train', dict(VFS_SYNCBN_P2P='1', VFS_FIN_XCHG='1', VFS_TEST_STEPS='3'))
    coll = _run_two(tmp_path, 'coll2', 'train', dict(VFS_SYNCBN_P2P='0', VFS_TEST_STEPS='3'))
    assert int(fold[0]['p2p_active']) == 1 and int(fold[0]['p2p_exchanges']) > 0
    assert int(fold[0]['p2p_exchanges']) == int(fold[1]['p2p_exchanges'])
    n, worst = 0, 0.0
    for k in coll[0].files:
        if k.startswith(('param/', 'grad/', 'buf/')):
            assert np.array_equal(fold[0][k], fold[1][k]), k
        if k.startswith(('param/', 'buf/')):
            a, b = fold[0][k].astype(np.float64), coll[0][k].astype(np.float64)
            worst = max(worst, float(np.abs(a - b).max() / max(1e-12, np.abs(b).max())))
            n += 1
    assert n > 50 and worst < 1e-5, worst
    for k in coll[0].files:
        if k.startswith('log/'):
            assert abs(float(fold[0][k]) - float(coll[0][k])) < 1e-5 * max(1.0, abs(float(coll[0][k]))), k


@pytest.mark.gpu
def test_two_processes_one_gpu_poisoned_step_is_skipped_on_every_rank(gpu_backend, tmp_path):
    """ONE rank's SyncBN exchange times out in the last of two steps (its error word set, as vfs_p2p.h does): the word is
    MAX-reduced behind the gradient buckets, so BOTH ranks' sgd_step leaves the weights untouched and both end with the word set
    (round 4 advisor finding: only the failing rank skipped, the healthy rank - the checkpoint writer - applied its peer's
    gradients)"""
    r = _run_two(tmp_path, 'poison', 'train', dict(VFS_SYNCBN_P2P='1', VFS_TEST_STEPS='2', VFS_TEST_POISON_RANK='1'))
    for k in range(2):
        assert int(r[k]['p2p_active']) == 1
        assert int(r[k]['poison_unchanged']) == 1, f'rank {k} applied a poisoned step'
        assert int(r[k]['poison_word']) != 0
    for key in r[0].files:
        if key.startswith('param/'):
            assert np.array_equal(r[0][key], r[1][key]), key
