"""`python bench.py --gpus N` starts its own ranks (the driver's command form died on an assert unless something
wrapped it in torch.distributed.run).  Reference behaviour: tools/dist_train.sh:7-9 launches the N ranks, apis/train.py:58-66
wraps the model for them.
  * CPU: without a visible GPU the launcher refuses with a clear message and a non-zero code (no hang, no traceback from a rank);
    a WORLD_SIZE that contradicts --gpus is refused as well;
  * GPU (one device on the box): `--gpus 2` runs two ranks in shared-device mode on cuda:0 - gradients through gloo, SyncBN
    statistics through the IPC windows - and rank 0 prints ONE JSON line that says how the ranks talked."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')
SMALL = ['--model', 'r18', '--batch', '4', '--size', '64', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-davis']


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(kw)
    return env


def test_self_launch_without_gpu_is_refused_cleanly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2'] + SMALL, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert 'no GPU visible' in r.stderr and 'Traceback' not in r.stderr
    assert r.stdout.strip() == ''


def test_world_size_mismatch_is_refused_cleanly():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2'] + SMALL, env=_env(WORLD_SIZE='3', RANK='0', LOCAL_RANK='0'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'WORLD_SIZE=3' in r.stderr and 'Traceback' not in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_starts_its_own_ranks(gpu_backend):
    import torch
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2'] + SMALL, env=_env(VFS_P2P_SPIN=str(1 << 26)), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['steps'] == 3 and res['value'] > 0
    d = res['distributed']
    assert d['ranks'] == 2
    if torch.cuda.device_count() < 2:
        assert d['shared_device'] is True and 'gloo' in d['collective_backend']
        assert 'SHARING one device' in res['config']['parallelism']
        assert 'SHARED-DEVICE' in r.stderr
    else:
        assert d['shared_device'] is False and 'RCCL' in d['collective_backend']
    assert d['syncbn_statistics'].startswith('IPC-window') and d['syncbn_exchanges'] > 0
    assert d['allreduce_ms_per_bucket']['median'] > 0
    assert res['loss'] == res['loss'] and abs(res['loss']) < 1e3      # finite
