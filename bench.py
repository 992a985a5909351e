#!/usr/bin/env python3
"""Benchmark of the VFS training hot path on MI355X: frame-pairs/s of the SimSiam train step
(forward_train + backward + SGD) on synthetic 256x256 clips.

  python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one per-GPU batch: imgs [32, 2, 3, T, 256, 256].
Default = the configuration BASELINE.json's metric is quoted on ("frame-pairs/sec (train) R50 256^2 at
1/2/4/8 GPUs": configs[2], ResNet-50 r5_1xNx2, T=1 -> 32 frame-pairs per GPU per step; it fits one GPU);
--model r18: configs[1], ResNet-18 r2_1xNx8, T=4 -> 128 frame-pairs.  Inputs are resident in HBM before
the timed region.  Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E, same guide


def log(*a):
    print(f'[bench {time.strftime("%H:%M:%S")}]', *a, file=sys.stderr, flush=True)


def cpu_baseline_subprocess(depth, size, threads, timeout=240):
    """run cpu_baseline() in a child so a pathological host (thread oversubscription) cannot
    hang the benchmark; returns the dict or a null entry with the reason"""
    import subprocess
    code = (f'import sys, json; sys.path.insert(0, {REPO!r}); import bench; '
            f'print("CPUBASE " + json.dumps(bench.cpu_baseline({depth}, {size}, {threads})))')
    try:
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=timeout)
        for line in out.stdout.splitlines():
            if line.startswith('CPUBASE '):
                return json.loads(line[8:])
        return dict(value=None, unit='frame-pairs/s', cores=threads, kind='port', sample='failed: ' + out.stderr[-300:])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit='frame-pairs/s', cores=threads, kind='port', sample=f'timed out after {timeout}s')


def cpu_baseline(depth, size, threads):
    """The oracle (CPU restatement of the reference's PyTorch path, fp32) timed on the host cores
    on a bounded sample of the same workload."""
    from oracle import vfs_oracle as O
    torch.set_num_threads(threads)
    T = 4 if depth == 18 else 1
    B = 2 if depth == 18 else 4
    model = O.build_tracker(depth).train()
    params = [p for p in model.parameters()]
    bufs = [None] * len(params)
    imgs = torch.randn(B, 2, 3, T, size, size, generator=torch.Generator().manual_seed(0))

    def step():
        for p in params:
            p.grad = None
        loss, _ = O.parse_losses(model.forward_train(imgs))
        loss.backward()
        with torch.no_grad():
            O.sgd_step(params, [p.grad for p in params], bufs, lr=0.05)
    step()
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 10 and n < 20):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return dict(value=B * T / dt, unit='frame-pairs/s', cores=threads, kind='port',
                sample=f'oracle (fp32 torch-CPU restatement) R{depth} train step, imgs [{B},2,3,{T},{size},{size}], '
                       f'1 warm-up + {n} timed steps, {dt * 1e3:.0f} ms/step')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--model', default='r50', choices=['r18', 'r50'])
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--batch', type=int, default=32, help='videos per GPU (configs: videos_per_gpu=32)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    force_coll = os.environ.get('VFS_FORCE_COLLECTIVES') == '1'   # 1-rank RCCL group: exercises the collective calls
    if world > 1 or force_coll:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)

    import vfs_amd
    from vfs_amd.engine import shared_engine
    depth = 18 if args.model == 'r18' else 50
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_r{depth}.py'))
    torch.manual_seed(0)
    model = vfs_amd.build_model(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev).train()
    flat = model.flatten_parameters()
    if world > 1:
        dist.broadcast(flat['params'], 0)
    opt = vfs_amd.build_optimizer(model, cfg.optimizer)
    T = int(cfg.clip_len)
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    imgs = torch.randn(B, 2, 3, T, args.size, args.size, device=dev, generator=g)
    batch = dict(imgs=imgs, label=torch.zeros(B, 1, device=dev))
    eng = shared_engine()

    def step():
        out = model.train_step(batch, opt)
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
        return out

    # initialisation (the analogue of a graph capture): one eager pass settles buffers / workspaces / packed weights, the
    # next one records the forward and backward command tapes.  No optimizer step; not part of warm-up or timing.
    for _ in range(2):
        o = model.train_step(batch, opt)
        opt.zero_grad()
        o['loss'].backward()
    torch.cuda.synchronize()
    log('launch chains recorded')

    for i in range(args.warmup):
        t_w = time.perf_counter()
        out = step()
        torch.cuda.synchronize()
        log(f'warmup step {i}: {(time.perf_counter() - t_w) * 1e3:.1f} ms, loss {out["log_vars"]["loss"]:.4f}')

    def timed(nsteps):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            o = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        tm = torch.tensor([d], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm.item()), o

    # ---- the timed region: EXACTLY args.steps steps (single process: hipGraph replay of the step)
    dt, out = timed(args.steps)
    log(f'{args.steps} timed steps: {dt / args.steps * 1e3:.2f} ms/step')
    # ---- roofline: per-kernel HIP events need eager launches (a graph replay is one launch), so the
    # same number of steps is repeated eagerly right after the timed region with an event pair around
    # every conv launch (events pre-created; only hipEventRecord is added)
    prof, dt_prof = None, None
    if not args.no_roofline:
        # one stream for this leg: with the weight-gradient kernels running concurrently on the side
        # stream a per-kernel event pair would time two overlapping kernels, not one
        side_prev = os.environ.get('VFS_SIDE_STREAM')
        os.environ['VFS_SIDE_STREAM'] = '0'
        eng.prof = []
        step()
        torch.cuda.synchronize()
        per_step = len(eng.prof)
        eng.prof_pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * per_step * args.steps + 16)]
        eng.prof = []
        dt_prof, _ = timed(args.steps)
        log(f'{args.steps} eager steps with per-kernel events: {dt_prof / args.steps * 1e3:.2f} ms/step')
        prof, eng.prof = eng.prof, None
        if side_prev is None:
            os.environ.pop('VFS_SIDE_STREAM')
        else:
            os.environ['VFS_SIDE_STREAM'] = side_prev

    pairs_per_step = B * T * world
    res = {
        'metric': f'frame-pairs/sec (train) R{depth} {args.size}\u00b2', 'value': pairs_per_step * args.steps / dt, 'unit': 'frame-pairs/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': f'ResNet-{depth} SimSiam (VFS) forward_train+backward+SGD, imgs [{B},2,3,{T},{args.size},'
                               f'{args.size}] per GPU (configs[{1 if depth == 18 else 2}] shape), SyncBN, fp32 master weights',
                   'frame_pairs_per_step': pairs_per_step, 'parallelism': f'dp{world}'},
        'loss': out['log_vars']['loss'],
        'launch_mode': ('hipGraph replay (forward chain + backward chain)' if (world == 1 and os.environ.get('VFS_GRAPHS', '0') == '1')
                        else 'command-tape replay (recorded C-ABI calls + stream waits / collectives)' if os.environ.get('VFS_TAPE', '1') == '1'
                        else 'eager'),
    }
    if rank == 0 and prof:
        agg = {}
        for kind, flops, e0, e1, nbytes in prof:
            a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
            a[3] += nbytes
        kind = max(agg, key=lambda k: agg[k][1])
        fl, tm, cnt, nb = agg[kind]
        ach = fl / tm / 1e12
        traffic = None   # HBM bytes per launch from the committed PMC passes (tools/gpu_pmc.sh), if present
        tpath = os.path.join(REPO, 'profiles', f'r01_traffic_{args.model}.json')
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get('classes', {}).get(kind, {}).get('hbm_bytes_per_launch')
        # which roof binds the class: time at the dense bf16 MFMA peak for the algorithmic FLOP vs time at the
        # HBM peak for the algorithmic bytes (every operand of a launch once; Engine.timed) - peaks from
        # MI355X_MICROARCH.md: 2.5 PFLOP/s, 8 TB/s.  `traffic` is what the PMC passes actually counted.
        t_launch = tm / cnt
        t_mfma = fl / cnt / (PEAK_BF16_TFLOPS * 1e12)
        t_hbm = nb / cnt / (PEAK_HBM_GBS * 1e9)
        extra = {'algorithmic_flop_per_launch': fl / cnt, 'algorithmic_bytes_per_launch': nb / cnt, 'launches': cnt,
                 'avg_launch_ms': t_launch * 1e3, 'time_share_of_step': tm / dt_prof,
                 'mfma': {'achieved_TFLOP/s': ach, 'frac': ach / PEAK_BF16_TFLOPS},
                 'hbm': {'achieved_GB/s': nb / tm / 1e9, 'frac': nb / tm / 1e9 / PEAK_HBM_GBS,
                         'measured_traffic_GB/s': (traffic or 0.0) / t_launch / 1e9},
                 'measured_over': f'{args.steps} eager single-stream steps right after the timed region, {dt_prof / args.steps * 1e3:.2f} ms/step',
                 'others': {k: {'TFLOP/s': v[0] / v[1] / 1e12, 'GB/s': v[3] / v[1] / 1e9, 'time_share_of_step': v[1] / dt_prof}
                            for k, v in agg.items() if k != kind}}
        if t_hbm > t_mfma:
            res['roofline'] = {'kernel': kind, 'bound': 'hbm', 'achieved': extra['hbm']['achieved_GB/s'], 'peak': PEAK_HBM_GBS,
                               'unit': 'GB/s', 'frac': extra['hbm']['frac'], 'traffic': traffic, **extra}
        else:
            res['roofline'] = {'kernel': kind, 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / PEAK_BF16_TFLOPS, 'traffic': traffic, **extra}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('timing the CPU oracle (bounded sample) ...')
        res['cpu_baseline'] = cpu_baseline_subprocess(depth, args.size, min(os.cpu_count() or 1, 64))
    if rank == 0:
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
