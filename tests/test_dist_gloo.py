"""Data-parallel path on the CPU: 2 gloo ranks, each running the fused step (kernels through the
fiber emulator) on half of the batch, must reproduce the single-process full-batch step:
SyncBN statistics all-reduced per view, gradients mean-all-reduced in buckets, log vars averaged
(reference: MMDistributedDataParallel + SyncBN + _parse_losses, apis/train.py:58-66,
trackers/base.py:103-108)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, 'tests', 'dist_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world', [2, 4])
def test_ranks_equal_one_process(tmp_path, world):
    """2 ranks (four videos each) and 4 ranks (four videos each) against the single process on the whole batch.
    The forward pass is bit-identical in every partition (same statistics rows, fp64 sums).  In the backward pass the BatchNorm sums
    of a rank are fp32 partial sums over ITS rows: they differ from the single process's in the last fp32 bit (1e-7), which flips a
    bf16 rounding of one dx element in the head now and then - and each flip fans out through the 3x3 convolutions below it
    (measured at 4 ranks: 1, 2, 5, 8, 32, 81, ... 19 009 differing elements layer by layer down to the stem).  With 2 ranks x 4
    videos no flip happens on this input and the gradients agree to the fp32 summation order (2e-6); with 4 ranks they agree to
    bf16 noise (2e-3 at the stem) - the bar there is 2e-2, and what the test pins is the plumbing: world = 4 bucket / SyncBN
    arithmetic, replicas bit-identical to each other, loss equal to the single process's."""
    from tests.emu_util import emu_lib
    emu_lib()                                  # build once, before the ranks race for it
    single = str(tmp_path / 'single.npz')
    batch = str(4 * world)
    env0 = dict(os.environ, WORLD_SIZE='1', RANK='0', VFS_TEST_BATCH=batch)
    subprocess.run([sys.executable, WORKER, single], check=True, env=env0, timeout=600)
    port = str(_free_port())
    procs, outs = [], []
    for r in range(world):
        o = str(tmp_path / f'rank{r}.npz')
        outs.append(o)
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=port, VFS_TEST_BATCH=batch)
        procs.append(subprocess.Popen([sys.executable, WORKER, o], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    s = np.load(single)
    r0, r1 = np.load(outs[0]), np.load(outs[-1])

    def l2(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))

    worst = []
    for k in s.files:
        if k.startswith('log/'):
            assert abs(float(r0[k]) - float(s[k])) < 2e-3 and abs(float(r0[k]) - float(r1[k])) < 1e-6, k
        elif k.startswith('buf/'):
            assert l2(r0[k], s[k]) < 1e-4 and np.array_equal(r0[k], r1[k]), k
        elif k.startswith('grad/'):
            assert np.array_equal(r0[k], r1[k]), k             # both ranks hold the reduced gradient
            if np.linalg.norm(s[k]) > 1e-3:
                worst.append((l2(r0[k], s[k]), k))
                assert l2(r0[k], s[k]) < (1e-4 if world == 2 else 2e-2), (k, l2(r0[k], s[k]))       # measured: <= 2e-6 / 5e-3
        elif k.startswith('param/'):
            assert np.array_equal(r0[k], r1[k]), k             # replicas stay in lock-step after SGD
    print(f'largest relative L2 gradient differences {world} ranks vs 1 process:', sorted(worst, reverse=True)[:5])


def _run_ranks(tmp_path, tag, extra_env):
    port = str(_free_port())
    procs, outs = [], []
    for r in range(2):
        o = str(tmp_path / f'{tag}_rank{r}.npz')
        outs.append(o)
        env = dict(os.environ, WORLD_SIZE='2', RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=port, **extra_env)
        procs.append(subprocess.Popen([sys.executable, WORKER, o], env=env))
    for p in procs:
        assert p.wait(timeout=1500) == 0
    return [np.load(o) for o in outs]


def test_command_tape_replay_equals_eager_with_collectives(tmp_path):
    """4 data-parallel steps on 2 gloo ranks: step 1 eager, step 2 recorded, steps 3-4 replayed from the command tapes
    (C-ABI calls + SyncBN / gradient all-reduces) must equal the plain eager run bit for bit"""
    from tests.emu_util import emu_lib
    emu_lib()
    tape = _run_ranks(tmp_path, 'tape', dict(VFS_TEST_STEPS='4', VFS_TAPE='1'))
    eager = _run_ranks(tmp_path, 'eager', dict(VFS_TEST_STEPS='4', VFS_TAPE='0'))
    for k in eager[0].files:
        assert np.array_equal(tape[0][k], eager[0][k]), k
        if k.startswith(('param/', 'grad/', 'buf/')):
            assert np.array_equal(tape[0][k], tape[1][k]), k


def test_bf16_gradient_buckets(tmp_path):
    """VFS_GRAD_BF16=1 (opt-in): the gradient arena is reduced in bf16 buckets - the mean gradient both ranks end up with equals the
    fp32-bucket result to bf16 rounding, and the replicas still hold identical bits"""
    from tests.emu_util import emu_lib
    emu_lib()
    f32 = _run_ranks(tmp_path, 'f32', dict(VFS_TEST_STEPS='1'))
    b16 = _run_ranks(tmp_path, 'b16', dict(VFS_TEST_STEPS='1', VFS_GRAD_BF16='1'))
    n = 0
    for k in f32[0].files:
        if k.startswith('grad/'):
            assert np.array_equal(b16[0][k], b16[1][k]), k
            a, b = b16[0][k].astype(np.float64), f32[0][k].astype(np.float64)
            if np.linalg.norm(b) > 1e-3:      # (the per-rank shares of some gradients cancel: compare norms, as the fp32 test does)
                assert np.linalg.norm(a - b) <= 2e-2 * np.linalg.norm(b), (k, np.linalg.norm(a - b) / np.linalg.norm(b))
                n += 1
    assert n > 10
