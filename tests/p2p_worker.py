"""Worker of tests/test_p2p.py: one of TWO processes sharing cuda:0 (gloo for rendezvous / reference sums).
usage: p2p_worker.py protocol|train OUT.npz   (env RANK / WORLD_SIZE / MASTER_*)"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def protocol(rank, world, out):
    from vfs_amd._lib import get_lib
    from vfs_amd.p2p import P2PExchange
    dev = torch.device('cuda:0')
    lib = get_lib()
    x = P2PExchange(lib, dev, spin_limit=1 << 24)
    assert x.self_test()
    g = torch.Generator().manual_seed(100 + rank)
    sizes = [1, 7, 256, 1000, 8192, 33, 4096, 2, 513, 64]
    mism, N = 0, 300
    for k in range(N):
        n = sizes[k % len(sizes)]
        mine = torch.randn(n, generator=g, dtype=torch.float64)
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        want = both[0].clone()
        for r in range(1, world):
            want = want + both[r]
        t = mine.to(dev)
        x.allreduce(lib, t, None)
        mism += int(not torch.equal(t.cpu(), want))
    torch.cuda.synchronize()
    # latency: back-to-back exchanges of a 4 KB payload (what a 256-channel BatchNorm sends for two views)
    t = torch.zeros(512, dtype=torch.float64, device=dev)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        x.allreduce(lib, t, None)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 200 * 1e6
    np.savez(out, exchanges=N, mismatches=mism, error_word=int(x.failed()), us_per_exchange=us)
    dist.barrier()
    x.close()


def train(rank, world, out):
    import vfs_amd
    from vfs_amd import engine
    from tests.dist_worker import build
    dev = torch.device('cuda:0')
    eng = engine.Engine()
    engine.set_shared_engine(eng)
    model, imgs, cfg = build(8)
    model.to(dev)
    per = imgs.shape[0] // world
    local = imgs[rank * per:(rank + 1) * per].to(dev)
    opt = vfs_amd.build_optimizer(model, cfg.optimizer)
    poison = os.environ.get('VFS_TEST_POISON_RANK')
    before = None
    for step in range(int(os.environ.get('VFS_TEST_STEPS', '1'))):
        batch = local if step == 0 else (local * (1.0 + 0.25 * step)).contiguous()
        o = model.train_step(dict(imgs=batch, label=torch.zeros(per, 1, device=dev)), None)
        opt.zero_grad()
        if poison is not None and step == int(os.environ.get('VFS_TEST_STEPS', '1')) - 1:
            # the last step: ONE rank's exchange "timed out" (its error word is set, as vfs_p2p.h does) - no rank may update
            torch.cuda.synchronize()
            before = {n: p.detach().clone() for n, p in model.named_parameters()}
            if rank == int(poison):
                eng._p2p.state[1] = 1
        o['loss'].backward()
        opt.step()
    torch.cuda.synchronize()
    res = {'log/' + k: np.float64(v) for k, v in o['log_vars'].items()}
    for n, p in model.named_parameters():
        res['grad/' + n] = p.grad.detach().cpu().numpy().copy()
        res['param/' + n] = p.detach().cpu().numpy().copy()
    for n, b in model.named_buffers():
        if 'running' in n:
            res['buf/' + n] = b.cpu().numpy().copy()
    if before is not None:
        res['poison_unchanged'] = int(all(torch.equal(before[n], p.detach()) for n, p in model.named_parameters()))
        res['poison_word'] = int(eng._p2p.state[1].item())
    res['p2p_active'] = int(eng._p2p is not None)
    res['p2p_exchanges'] = int(eng._p2p.state[0].item()) if eng._p2p is not None else 0
    np.savez(out, **res)
    dist.barrier()


def _stage_cuda_allreduce_through_host():
    """RCCL refuses two ranks on one device, so this test's process group is gloo; its all-reduces of DEVICE tensors (gradient
    buckets, log values, and - on the comparison run - the SyncBN statistics) are staged through the host.  Test plumbing only."""
    orig = dist.all_reduce

    def staged(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if t.is_cuda:
            h = t.detach().cpu()
            orig(h, op=op, group=group)
            t.copy_(h)
            return None
        return orig(t, op=op, group=group, async_op=async_op)
    dist.all_reduce = staged


if __name__ == '__main__':
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _stage_cuda_allreduce_through_host()
    {'protocol': protocol, 'train': train}[sys.argv[1]](rank, world, sys.argv[2])
    dist.destroy_process_group()
