#!/usr/bin/env python3
"""Benchmark of the VFS training hot path on MI355X: frame-pairs/s of the SimSiam train step
(forward_train + backward + SGD) on synthetic 256x256 clips.

  python bench.py --gpus N --steps K --warmup W

N > 1: when the ranks are not there yet (no WORLD_SIZE in the environment) bench.py STARTS THEM ITSELF - it re-executes this
command under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL),
the reference's tools/dist_train.sh:7-9 / apis/train.py:58-66 in one call; launched BY torch.distributed.run (the driver's
form) it is a rank.  On a box with fewer than N devices the N ranks share cuda:0 ("shared-device" mode: gradients through gloo,
SyncBN statistics through the IPC windows - a functional check of the N > 1 code path, NOT a scaling number; the JSON says so).

One "step" = one pass of the hot path over one per-GPU batch: imgs [32, 2, 3, T, 256, 256].
Default = the configuration BASELINE.json's metric is quoted on ("frame-pairs/sec (train) R50 256^2 at
1/2/4/8 GPUs": configs[2], ResNet-50 r5_1xNx2, T=1 -> 32 frame-pairs per GPU per step; it fits one GPU);
--model r18: configs[1], ResNet-18 r2_1xNx8, T=4 -> 128 frame-pairs.  Inputs are resident in HBM before
the timed region.  Prints ONE JSON line on rank 0.

  python bench.py --workload davis [--model r50|r18] [--precision fp32|bf16] --steps K --warmup W

BASELINE.json configs[3] (the second headline metric's workload): DAVIS label propagation on a synthetic
480x854 clip with the reference's test-time settings; one "step" = one propagated frame (backbone of the frame,
L2 normalisation, masked attention over first + 20 preceding frames, upsample / min-max / argmax).  Same JSON
schema (`roofline` on the in-mask affinity FLOP, `cpu_baseline` = the C oracle per frame).  N > 1: replicas."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
os.environ.setdefault('VFS_GC_FREEZE', '1')     # this process is the benchmark's own: see trackers._freeze_gc_once

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3     # fp32-input MFMA (v_mfma_f32_32x32x2_f32) = the fp32 vector peak, same guide
PEAK_HBM_GBS = 8000.0       # HBM3E, same guide
# algorithmic work per frame-pair (3 x forward FLOP; ideal-fusion activation traffic, bf16)
WORK_PER_PAIR = {(50, 256): (64.2e9, 232e6), (50, 224): (49.2e9, 178e6), (50, 512): (256.4e9, 929e6),
                 (18, 256): (28.4e9, 52e6), (18, 224): (21.8e9, 40e6)}


def log(*a):
    print(f'[bench {time.strftime("%H:%M:%S")}]', *a, file=sys.stderr, flush=True)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """--gpus N > 1 without ranks: start N ranks of this very command (tools/dist_train.sh:7-9 of the reference does the same with
    torch.distributed.launch).  Returns the launcher's exit code; rank 0's JSON line reaches stdout through the launcher."""
    import subprocess
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL and the SyncBN windows between processes
    if ndev < 1:
        log(f'--gpus {args.gpus}: no GPU visible to this process')
        return 2
    if ndev < args.gpus:
        log(f'--gpus {args.gpus} needs {args.gpus} devices, this box has {ndev}: running the {args.gpus} ranks in SHARED-DEVICE mode on '
            'cuda:0 (gloo gradients staged through the host, SyncBN statistics through the IPC windows).  This exercises the N > 1 '
            'code path; it is NOT a scaling measurement.')
        env['VFS_BENCH_SHARED_DEVICE'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    log('starting the ranks:', ' '.join(cmd))
    return subprocess.call(cmd, env=env)


def stage_cuda_collectives_through_host():
    """shared-device mode only: RCCL refuses two ranks on one device, so the process group is gloo and its collectives on DEVICE
    tensors (gradient buckets, the log values, broadcast of the initial weights) are staged through the host"""
    orig_ar, orig_bc = dist.all_reduce, dist.broadcast

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if t.is_cuda:
            h = t.detach().cpu()
            orig_ar(h, op=op, group=group)
            t.copy_(h)
            return None
        return orig_ar(t, op=op, group=group, async_op=async_op)

    def broadcast(t, src=0, group=None, async_op=False):
        if t.is_cuda:
            h = t.detach().cpu()
            orig_bc(h, src, group=group)
            t.copy_(h)
            return None
        return orig_bc(t, src, group=group, async_op=async_op)
    dist.all_reduce, dist.broadcast = all_reduce, broadcast


# kernel name (prefix after "void ") -> bench.py's family label, for the committed rocprofv3 --stats CSV of the TIMED schedule
KERNEL_FAMILY = (('conv_igemm_kernel', 'conv_igemm'), ('conv_skinny_kernel', 'conv_igemm'), ('conv_pw_kernel', 'conv_igemm'), ('linear_bn_act_kernel', 'conv_igemm'), ('conv3x3_halo_kernel', 'conv3x3_halo'), ('stem_fwd_direct_kernel', 'stem_fwd'),
                 ('conv_wgrad_kernel', 'conv_wgrad'), ('conv_wgrad_ring_kernel', 'conv_wgrad'), ('conv3x3_wgrad_halo_kernel', 'conv3x3_wgrad_halo'), ('stem_wgrad_fused_kernel', 'stem_wgrad'),
                 ('wgrad_reduce_', 'wgrad_reduce'), ('bn_act_kernel', 'bn_act'), ('bn_bwd_apply_kernel', 'bn_bwd_apply'),
                 ('bn_bwd_reduce_kernel', 'bn_bwd_reduce'), ('stem_pool_bn_bwd_reduce', 'bn_bwd_reduce'), ('bn_reduce_', 'bn_stats'),
                 ('bn_stats_raw', 'bn_stats'), ('bn_finalize', 'bn_stats'), ('bn_relu_maxpool_kernel', 'bn_relu_maxpool'),
                 ('pack_weights_kernel', 'pack_weights'), ('sgd_kernel', 'sgd'), ('labelprop_f32', 'labelprop_f32'),
                 ('lp2_', 'labelprop_2pass'), ('conv_f32_kernel', 'conv_f32'), ('conv_f32_db_kernel', 'conv_f32'), ('seg_minmax_exact', 'seg_postprocess'),
                 ('seg_argmax_exact', 'seg_postprocess'))


def rocprof_families(tag):
    """per-family kernel time of the TIMED schedule (command-tape replay, weight gradients on the side stream) from the committed
    `rocprofv3 --kernel-trace --stats` summary of this command: profiles/<round>_<tag>_kernel_stats.csv + .meta.json
    ({"passes": steps the profiled run executed, ...}; tools/gpu_evidence.sh writes both).  None when no summary is committed."""
    import csv
    import glob
    cands = sorted(glob.glob(os.path.join(REPO, 'profiles', f'r[0-9][0-9]_{tag}_kernel_stats.csv')))
    if not cands:
        return None
    path = cands[-1]
    meta_path = path[:-4] + '.meta.json'
    if not os.path.exists(meta_path):
        return None
    meta = json.load(open(meta_path))
    passes = float(meta['passes'])
    fams, total = {}, 0.0
    for r in csv.DictReader(open(path)):
        name = r['Name'].replace('void ', '')
        fam = next((f for k, f in KERNEL_FAMILY if name.startswith(k)), None)
        ns, calls = float(r['TotalDurationNs']), int(r['Calls'])
        total += ns
        if fam is None:
            continue
        a = fams.setdefault(fam, [0.0, 0])
        a[0] += ns
        a[1] += calls
    return {'source': os.path.relpath(path, REPO), 'passes': passes, 'command': meta.get('command'),
            'kernel_ms_per_pass_all': total / passes / 1e6,
            'families': {f: {'kernel_ms_per_pass': v[0] / passes / 1e6, 'launches_per_pass': v[1] / passes, 'avg_launch_us': v[0] / v[1] / 1e3}
                         for f, v in sorted(fams.items(), key=lambda kv: -kv[1][0])}}


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


def extra_train_leg(model, size, batch, steps, warmup, timeout=300):
    """one more training configuration of BASELINE.json (configs[1]: ResNet-18 T = 4 at 256^2, configs[4]: ResNet-50 at 512^2) through
    THIS script in a child process (its own engine and buffers, the GPU is idle meanwhile): the child's JSON line, cut down to what
    a reader of the headline line needs - value, ms_per_step, the step-level roofline and the dominant kernel family"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--model', model, '--size', str(size), '--batch', str(batch), '--steps', str(steps),
           '--warmup', str(warmup), '--no-cpu-baseline', '--no-davis', '--no-extra-legs']
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        line = next((ln for ln in reversed(out.stdout.splitlines()) if ln.startswith('{')), None)
        if line is None:
            return {'value': None, 'error': out.stderr[-300:]}
        r = json.loads(line)
        rf = r.get('roofline') or {}
        return {'metric': r['metric'], 'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'], 'steps': r['steps'],
                'warmup': r['warmup'], 'dtype': r['dtype'], 'config': r['config'], 'loss': r.get('loss'), 'step_roofline': r.get('step_roofline'),
                'roofline': {k: rf.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launches_per_step',
                                                    'avg_launch_ms', 'kernel_ms_per_step', 'traffic_source')} if rf else None,
                'command': ' '.join(['python', 'bench.py'] + cmd[2:])}
    except subprocess.TimeoutExpired:
        return {'value': None, 'error': f'timed out after {timeout}s'}


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


def davis_cpu_baseline(depth, threads):
    """The C oracle (oracle/exact_oracle.c: the reference's fp32 evaluation arithmetic) on the host cores, one frame of the
    same workload: ResNet stem..res4 of ONE 480x854 frame + ONE propagation step over 21 key frames + post-processing."""
    import numpy as np
    os.environ['OMP_NUM_THREADS'] = str(threads)
    from oracle import exact_oracle as X
    from oracle import vfs_oracle as O
    ref = O.ResNet(depth, strides=(1, 2, 1, 1), out_indices=(2,))
    O.fill_state_dict_(ref, seed=5)
    sd = ref.state_dict()
    H, W, h, w = 480, 854, 60, 107
    C = 256 if depth == 18 else 1024
    radius = 12 if depth == 18 else 18
    frame = np.random.RandomState(0).randn(1, 3, H, W).astype(np.float32)
    t0 = time.perf_counter()
    X.resnet_eval(sd, depth, frame, strides=(1, 2, 1, 1), out_indices=(2,))
    t_bb = time.perf_counter() - t0
    rs = np.random.RandomState(1)
    bank = X.l2norm_rows(rs.randn(22 * h * w, C).astype(np.float32)).reshape(22, h * w, C)
    sbank = rs.rand(22, h * w, 4).astype(np.float32)
    t0 = time.perf_counter()
    out = X.labelprop(bank, sbank, 21, [0] + list(range(1, 21)), h, w, radius, 10, 0.07)
    t_lp = time.perf_counter() - t0
    t0 = time.perf_counter()
    X.seg_postprocess(out, h, w, H, W)
    t_pp = time.perf_counter() - t0
    tot = t_bb + t_lp + t_pp
    return dict(value=1.0 / tot, unit='frames/s', cores=threads, kind='port',
                sample=f'C oracle (fp32, gcc -O3 + OpenMP, {threads} threads), ONE 480x854 frame of the same workload: R{depth} stem..res4 '
                       f'{t_bb:.2f} s + propagation over 21 key frames {t_lp:.2f} s + post-processing {t_pp:.2f} s')


def bench_davis(args, depth, dev, world, rank, steps=None, warmup=None):
    """BASELINE.json configs[3]: DAVIS-2017 label propagation, synthetic 480x854 clip, the reference's test-time config"""
    import numpy as np
    import vfs_amd
    from vfs_amd.engine import shared_engine
    from vfs_amd.labelprop import mask_pairs
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_r{depth}.py'))
    tc = vfs_amd.ConfigDict(cfg.test_cfg)
    tc['precision'] = args.precision
    bb = dict(cfg.model['backbone'])
    bb['out_indices'], bb['strides'] = tc['out_indices'], tc['strides']          # tools/test.py:129-133
    model = vfs_amd.build_model(dict(type='VanillaTracker', backbone=bb), train_cfg=None, test_cfg=tc)
    from vfs_amd.synthetic import synthetic_weights_      # no checkpoint on the box: deterministic non-degenerate weights
    synthetic_weights_(model, seed=5)
    model.to(dev).eval()
    K, Wm = (steps or args.steps), max(1, args.warmup if warmup is None else warmup)
    H, W = 480, 854
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    base = torch.randn(1, 1, 3, 1, H, W, device=dev, generator=g)           # a slowly drifting scene: propagation is not pure noise
    imgs = base + 0.15 * torch.randn(1, 1, 3, K + 1, H, W, device=dev, generator=g)
    yy, xx = np.mgrid[0:H, 0:W]
    seg = np.zeros((H, W), np.uint8)
    seg[(yy > 100) & (yy < 300) & (xx > 150) & (xx < 400)] = 1
    seg[(yy > 250) & (yy < 420) & (xx > 500) & (xx < 760)] = 2
    seg[(yy - 120) ** 2 + (xx - 650) ** 2 < 80 ** 2] = 3
    seg_t, meta = torch.from_numpy(seg)[None], [dict(original_shape=(H, W, 3))]
    eng = shared_engine()

    def run(clip):
        return model(clip, return_loss=False, ref_seg_map=seg_t, img_meta=meta)
    run(imgs[:, :, :, :Wm + 1])            # W untimed propagated frames (weight repack, kernels loaded)
    run(imgs)                              # + one untimed pass over the timed clip itself: the feature / label banks of a K + 1 frame clip
    torch.cuda.synchronize()               #   (0.8 GB for ResNet-50) are allocated and touched here, not inside the timed region
    log(f'{Wm} warm-up frames + one untimed pass done')

    def timed():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(imgs)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        tm = torch.tensor([d], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm.item()), out
    dt, out = timed()
    log(f'{K} propagated frames: {dt / K * 1e3:.3f} ms/frame')
    # steady state: every frame from the 21st on propagates from the full key window (first + 20 preceding frames); the first 20
    # have fewer key frames (and cold thresholds).  steady = (whole clip - its first 21 frames) / the frames in between
    steady = None
    if K > 24:
        head = imgs[:, :, :, :21].contiguous()
        run(head)
        torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(head)
        torch.cuda.synchronize()
        dt_head = time.perf_counter() - t0
        steady = (dt - dt_head) / (K - 20) * 1e3
        log(f'first 20 propagated frames: {dt_head / 20 * 1e3:.3f} ms/frame; steady state (21 key frames): {steady:.3f} ms/frame')
    C = 256 if depth == 18 else 1024
    radius = int(tc['neighbor_range']) // 2
    res = {'metric': f'DAVIS label propagation frames/sec (R{depth} res4, 480x854, first + 20 preceding key frames)',
           'value': world * K / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': K, 'warmup': Wm, 'ms_per_step': dt / K * 1e3,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.precision == 'fp32' else 'bf16',
           'data': 'synthetic',
           'config': {'workload': f'VanillaTracker.forward_test (BASELINE configs[3]): ResNet-{depth} stem..res4 at stride 8 + masked '
                                  f'attention (top-10, tau 0.07, radius {radius}, first + 20 preceding frames) + upsample/min-max/argmax, '
                                  f'one {K + 1}-frame 480x854 clip per GPU, {args.precision} evaluation path',
                      'frames_per_clip': K + 1, 'parallelism': f'replicas x{world}'},
           'steady_state_ms_per_frame': steady,
           'labels_present': sorted(int(v) for v in np.unique(out[0][-1]))}
    if rank == 0 and not args.no_roofline:
        eng.prof = []
        run(imgs)
        torch.cuda.synchronize()
        prof, eng.prof = eng.prof, None
        agg = {}
        for kind, flops, e0, e1, nbytes in prof:
            a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
            a[3] += nbytes
        tot = sum(v[1] for v in agg.values())
        peak = PEAK_F32_TFLOPS if args.precision == 'fp32' else PEAK_BF16_TFLOPS

        # HBM bytes per launch from the committed PMC passes of this workload (tools/gpu_pmc.sh <model> davis + make_traffic_json.py)
        tclasses, tsource = {}, None
        for tag in ('r06', 'r05', 'r04', 'r03'):
            tpath = os.path.join(REPO, 'profiles', f'{tag}_traffic_davis_{args.model}.json')
            if args.precision == 'fp32' and os.path.exists(tpath):
                tclasses, tsource = json.load(open(tpath)).get('classes', {}), os.path.relpath(tpath, REPO)
                break

        def family(kind):
            fl, tm, cnt, nb = agg[kind]
            # the two-pass label propagation scores on the bf16 matrix path (three products per candidate): its roof is the bf16 peak
            fpeak = PEAK_BF16_TFLOPS if kind == 'labelprop_2pass' else peak
            # (the two-pass label propagation is a contraction: priced against the matrix roof whichever way the byte model falls)
            hbm_bound = nb / (PEAK_HBM_GBS * 1e9) >= fl / (fpeak * 1e12) and kind != 'labelprop_2pass'
            ach, pk, unit = (nb / tm / 1e9, PEAK_HBM_GBS, 'GB/s') if hbm_bound else (fl / tm / 1e12, fpeak, 'TFLOP/s')
            tr = tclasses.get(kind, {}).get('hbm_bytes_per_launch')
            ex = {'executed_frac': 3.0 * ach / pk, 'executed_TFLOP/s': 3.0 * ach} if (kind == 'labelprop_2pass' and not hbm_bound) else {}
            return {'kernel': kind, 'bound': 'hbm' if hbm_bound else 'mfma', 'achieved': ach, 'peak': pk, 'unit': unit, 'frac': ach / pk, **ex,
                    'traffic': tr, 'traffic_ratio': (tr / (nb / cnt)) if (tr and nb) else None, 'launches': cnt, 'avg_launch_ms': tm / cnt * 1e3, 'time_share_of_kernels': tm / tot,
                    'algorithmic_flop_per_launch': fl / cnt, 'algorithmic_bytes_per_launch': nb / cnt}
        kind = max(agg, key=lambda k: agg[k][1])
        res['roofline'] = family(kind)
        res['roofline']['traffic_source'] = tsource
        res['roofline']['note'] = ('FLOP = the affinity INSIDE the circular mask only (2 * C * in-mask (query, key) pairs per key frame; the '
                                   'dense T*HW x HW product the reference executes is not counted); labelprop_2pass (csrc/labelprop2.hip, same '
                                   'bits as the dense fp32 kernel): frac counts ONE product per in-mask pair (SURVEY 8(d)) against the dense bf16 MFMA peak '
                                   '(2.5 PFLOP/s); executed_frac = the THREE bf16 products per pair the kernel runs for it (hi.hi + hi.lo + lo.hi of the split '
                                   'bank) - it EXECUTES ~2x these again (the window union of an '
                                   '8 x 8 query tile against the in-mask keys of one query) with ONE wave per SIMD (the query tile fills the register '
                                   'file), whose in-order instruction stream is the limit (matrix pipe ~30 % busy; phase timers in MEASUREMENTS.md); '
                                   'other families: peak = dense '
                                   + ('fp32-input MFMA (157.3 TFLOP/s)' if args.precision == 'fp32' else 'bf16 MFMA (2.5 PFLOP/s)'))
        res['roofline']['families'] = [family(k) for k in sorted(agg, key=lambda k: -agg[k][1]) if k != kind and agg[k][1] / tot >= 0.03]
        res['roofline']['in_mask_pairs_per_key_frame'] = mask_pairs(60, 107, radius)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('timing the C oracle on one frame ...')
        import subprocess
        threads = min(os.cpu_count() or 1, 64)
        code = (f'import sys, json; sys.path.insert(0, {REPO!r}); import bench; '
                f'print("CPUBASE " + json.dumps(bench.davis_cpu_baseline({depth}, {threads})))')
        try:
            o = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=400)
            line = [ln for ln in o.stdout.splitlines() if ln.startswith('CPUBASE ')]
            res['cpu_baseline'] = json.loads(line[0][8:]) if line else dict(value=None, unit='frames/s', cores=threads, kind='port',
                                                                            sample='failed: ' + o.stderr[-300:])
        except subprocess.TimeoutExpired:
            res['cpu_baseline'] = dict(value=None, unit='frames/s', cores=threads, kind='port', sample='timed out after 400 s')
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--model', default='r50', choices=['r18', 'r50'])
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--batch', type=int, default=32, help='videos per GPU (configs: videos_per_gpu=32)')
    ap.add_argument('--workload', default='train', choices=['train', 'davis'])
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'], help='davis workload: evaluation precision')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-davis', action='store_true', help='train workload: skip the DAVIS leg appended to the JSON line (N = 1 only)')
    ap.add_argument('--davis-frames', type=int, default=49, help='propagated frames of the appended DAVIS leg (49: the T = 50 clip of BASELINE configs[3] / SURVEY 8(d) Cfg 4)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extra-legs', action='store_true', help='default line (ResNet-50, 256^2, N = 1): skip the appended ResNet-18 (configs[1]) and ResNet-50 512^2 (configs[4]) legs (--no-davis skips them too)')
    ap.add_argument('--min-seconds', type=float, default=0.0,
                    help='make the timed region at least this long: the number of timed steps is raised to ceil(min_seconds / step time) '
                         '(a lease-side GPU-busy monitor sampling every few seconds can then corroborate the figure); the JSON reports '
                         'the steps actually timed')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:      # no ranks yet: start them (the reference's tools/dist_train.sh in one call)
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        log(f'--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus {args.gpus}` without a launcher (it starts the ranks '
            f'itself), or launch {args.gpus} ranks')
        sys.exit(2)
    shared_dev = world > 1 and os.environ.get('VFS_BENCH_SHARED_DEVICE') == '1'      # more ranks than devices: every rank on cuda:0
    if shared_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    force_coll = os.environ.get('VFS_FORCE_COLLECTIVES') == '1'   # 1-rank RCCL group: exercises the collective calls
    if world > 1 or force_coll:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        if shared_dev:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            stage_cuda_collectives_through_host()
        else:
            dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)

    import vfs_amd
    from vfs_amd.engine import shared_engine
    if os.environ.get('VFS_TAPE_PROFILE') == '1':
        from vfs_amd._lib import Tape
        Tape.slowest = {}
    depth = 18 if args.model == 'r18' else 50
    if args.workload == 'davis':
        res = bench_davis(args, depth, dev, world, rank)
        if rank == 0:
            print(json.dumps(res))
        if dist.is_initialized():
            dist.destroy_process_group()
        return
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

    phases = [0.0] * 5 if os.environ.get('VFS_BENCH_PHASES') == '1' else None      # diagnostics: host wall time per call of a step

    def step():
        if phases is not None:
            t = [time.perf_counter()]
            out = model.train_step(batch, opt); t.append(time.perf_counter())
            opt.zero_grad(); t.append(time.perf_counter())
            out['loss'].backward(); t.append(time.perf_counter())
            opt.step(); t.append(time.perf_counter())
            out['log_vars']['loss']; t.append(time.perf_counter())
            for i in range(5):
                phases[i] += t[i + 1] - t[i]
            return out
        out = model.train_step(batch, opt)
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
        out['log_vars']['loss']      # the host reads the step's log values once per iteration, as mmcv's runner does (log_buffer.update)
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
    if phases is not None:
        phases[:] = [0.0] * 5
    import gc
    gc.collect()           # benchmark hygiene: no full collection of the warm-up's garbage inside the timed region
    gc.freeze()
    if args.min_seconds > 0:
        import math
        est, _ = timed(3)                                        # untimed estimate (max over the ranks: every rank gets the same count)
        args.steps = max(args.steps, int(math.ceil(args.min_seconds / (est / 3))))
        log(f'--min-seconds {args.min_seconds}: timing {args.steps} steps')
    dt, out = timed(args.steps)
    log(f'{args.steps} timed steps: {dt / args.steps * 1e3:.2f} ms/step')
    if phases is not None:
        log('host ms per step in train_step / zero_grad / backward / opt.step / log read (timed steps): ' +
            ' / '.join(f'{p / args.steps * 1e3:.2f}' for p in phases))
    from vfs_amd._lib import Tape
    log(f'launch chains were (re)recorded {getattr(model, "chain_resets", 0)} times, engine generation {eng.generation}, '
        f'host time in tape replay {Tape.host_seconds * 1e3:.1f} ms over the whole run')
    if Tape.slowest:
        top = sorted(Tape.slowest.items(), key=lambda kv: -kv[1][0])[:8]
        log('slowest tape ops (host ms total / calls): ' + ', '.join(f'{k}: {v[0] * 1e3:.1f}/{v[1]}' for k, v in top))
    # ---- roofline: per-kernel HIP events need eager launches (a graph replay is one launch), so the
    # same number of steps is repeated eagerly right after the timed region with an event pair around
    # every conv launch (events pre-created; only hipEventRecord is added)
    prof, dt_prof, prof_mode = None, None, None
    tape_mode = os.environ.get('VFS_TAPE', '1') == '1' and not (world == 1 and os.environ.get('VFS_GRAPHS', '0') == '1')
    if not args.no_roofline and tape_mode:
        # per-kernel durations of the schedule that was TIMED: the same command-tape replay (weight gradients on the side stream,
        # overlapping the dgrad chain) with every C-ABI launch bracketed by HIP events on the stream it is launched on
        # (vfs_amd/_lib.py Tape.timing); the one eager launch of the step (SGD) through the engine's own event pair
        from vfs_amd._lib import Tape
        Tape.timing, eng.prof = [], []
        step()
        torch.cuda.synchronize()
        per_step = len(Tape.timing) + len(eng.prof)
        Tape.event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * per_step * args.steps + 64)]
        eng.prof_pool = [torch.cuda.Event(enable_timing=True) for _ in range(4 * args.steps + 16)]
        Tape.timing, eng.prof = [], []
        dt_prof, _ = timed(args.steps)
        prof, Tape.timing, eng.prof, Tape.event_pool = Tape.timing + eng.prof, None, None, None
        prof_mode = (f'{args.steps} steps of the TIMED schedule (command-tape replay, two streams) right after the timed region, HIP events '
                     f'around every launch on its own stream: {dt_prof / args.steps * 1e3:.2f} ms/step with the events')
        log(f'{args.steps} replayed steps with per-kernel events: {dt_prof / args.steps * 1e3:.2f} ms/step')
    elif not args.no_roofline:
        # (hipGraph / eager launch modes) one stream for this leg: with the weight-gradient kernels running concurrently on the side
        # stream a per-kernel event pair on torch's current stream would time two overlapping kernels, not one
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
        prof_mode = (f'{args.steps} eager single-stream steps right after the timed region (FALLBACK: not the timed schedule), '
                     f'{dt_prof / args.steps * 1e3:.2f} ms/step (HIP events around every launch)')
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
                               f'{args.size}] per GPU (configs[{1 if depth == 18 else (4 if args.size == 512 else 2)}] shape), SyncBN, fp32 master weights',
                   'frame_pairs_per_step': pairs_per_step, 'parallelism': f'dp{world}'},
        'loss': out['log_vars']['loss'],
        'launch_mode': ('hipGraph replay (forward chain + backward chain)' if (world == 1 and os.environ.get('VFS_GRAPHS', '0') == '1')
                        else 'command-tape replay (recorded C-ABI calls + stream waits / collectives)' if os.environ.get('VFS_TAPE', '1') == '1'
                        else 'eager'),
    }
    if world > 1 or force_coll:
        # how the N ranks talk: collective backend, how the SyncBN statistics travelled, what a gradient bucket costs
        bucket = int(model.grad_bucket_bytes)
        g = flat['grads']
        n = min(g.numel(), max(1, bucket // 4))
        times = []
        for _ in range(5):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            dist.all_reduce(g[:n])
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        g.zero_()
        res['distributed'] = {
            'ranks': world, 'devices_visible': torch.cuda.device_count(),
            'collective_backend': 'gloo, device tensors staged through the host (shared-device mode)' if shared_dev else 'nccl (RCCL)',
            'shared_device': bool(shared_dev),
            'syncbn_statistics': 'IPC-window exchange (csrc/p2p.hip)' if eng._p2p is not None else 'collective-library all-reduce',
            'syncbn_exchanges': int(eng._p2p.state[0].item()) if eng._p2p is not None else 0,
            'gradient_bytes': int(g.numel()) * 4, 'gradient_bucket_bytes': bucket,
            'allreduce_ms_per_bucket': {'bytes': n * 4, 'min': min(times), 'median': sorted(times)[len(times) // 2],
                                        'note': 'blocking all-reduce of one bucket on an idle GPU, host-timed; in the step the buckets '
                                                'overlap the backward chain'}}
        if shared_dev:
            res['config']['parallelism'] = (f'dp{world}: {world} ranks SHARING one device (functional check of the N > 1 code path, '
                                            'not a scaling measurement)')
    if rank == 0 and prof:
        agg = {}
        for kind, flops, e0, e1, nbytes in prof:
            a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
            a[3] += nbytes
        # HBM bytes per launch of every family from the committed PMC passes (tools/gpu_pmc.sh + make_traffic_json.py:
        # separate --pmc FETCH_SIZE / WRITE_SIZE runs of this command, FETCH_SIZE doubled as the guide prescribes for gfx950)
        tclasses, tsource = {}, None
        for tag in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
            tpath = os.path.join(REPO, 'profiles', f'{tag}_traffic_{args.model}{"" if args.size == 256 else "_" + str(args.size)}.json')
            if os.path.exists(tpath):
                tclasses, tsource = json.load(open(tpath)).get('classes', {}), os.path.relpath(tpath, REPO)
                break

        def family(kind):
            """roofline entry of one kernel family: which roof binds (time at the dense MFMA peak for its algorithmic FLOP vs
            time at the HBM peak for its algorithmic bytes - every operand of a launch once), achieved / peak"""
            fl, tm, cnt, nb = agg[kind]
            tr = tclasses.get(kind, {}).get('hbm_bytes_per_launch')
            hbm_bound = nb / (PEAK_HBM_GBS * 1e9) >= fl / (PEAK_BF16_TFLOPS * 1e12)
            ach, peak, unit = (nb / tm / 1e9, PEAK_HBM_GBS, 'GB/s') if hbm_bound else (fl / tm / 1e12, PEAK_BF16_TFLOPS, 'TFLOP/s')
            return {'kernel': kind, 'bound': 'hbm' if hbm_bound else 'mfma', 'achieved': ach, 'peak': peak, 'unit': unit,
                    'frac': ach / peak, 'traffic': tr, 'launches_per_step': cnt / args.steps, 'avg_launch_ms': tm / cnt * 1e3,
                    'kernel_ms_per_step': tm / args.steps * 1e3,
                    # share of the REPORTED step and of the event-instrumented one (weight gradients on their own stream: families on
                    # the two streams overlap, so the shares of one step may add up to more than 1)
                    'time_share_of_step': tm / dt, 'time_share_of_profiled_step': tm / dt_prof,
                    'algorithmic_bytes_per_launch': nb / cnt,
                    'algorithmic_flop_per_launch': fl / cnt, 'GB/s': nb / tm / 1e9, 'TFLOP/s': fl / tm / 1e12}
        if os.environ.get('VFS_BENCH_SHAPES'):      # diagnostics: every distinct (family, work) launch of the eager leg with its rate
            shapes = {}
            for k, flops, e0, e1, nbytes in prof:
                a = shapes.setdefault((k, flops, nbytes), [0.0, 0])
                a[0] += e0.elapsed_time(e1) * 1e-3
                a[1] += 1
            with open(os.environ['VFS_BENCH_SHAPES'], 'w') as fsh:
                fsh.write(f'per-launch table, {prof_mode}; sorted by time per step\n')
                for (k, flops, nbytes), (tm, cnt) in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
                    fsh.write(f'{k:20s} {cnt / args.steps:5.1f}/step  avg {tm / cnt * 1e6:8.1f} us  per-step {tm / args.steps * 1e3:7.3f} ms  '
                              f'{nbytes / 1e6:8.1f} MB  {nbytes / (tm / cnt) / 1e12:5.2f} TB/s  {flops / 1e9:8.2f} GFLOP  {flops / (tm / cnt) / 1e12:7.1f} TFLOP/s\n')
        kind = max(agg, key=lambda k: agg[k][1])
        res['roofline'] = family(kind)
        res['roofline']['traffic_source'] = tsource
        res['roofline']['measured_over'] = prof_mode
        # every other family that takes >= 3 % of the step (BatchNorm / streaming kernels included), and what is left
        fams = sorted((k for k in agg if k != kind), key=lambda k: -agg[k][1])
        res['roofline']['families'] = [family(k) for k in fams if agg[k][1] / dt_prof >= 0.03]
        res['roofline']['small_families_time_share'] = sum(agg[k][1] for k in fams if agg[k][1] / dt_prof < 0.03) / dt_prof
        # launches without a family of their own (C-ABI calls outside Engine.timed: they appear as 'other:<entry point>')
        res['roofline']['unlabelled_time_share'] = sum(v[1] for k, v in agg.items() if k.startswith('other:')) / dt_prof
        res['roofline']['kernel_time_over_step'] = sum(v[1] for v in agg.values()) / dt_prof      # > 1: the two streams overlap
        # the same families on the schedule that was TIMED (tape replay, two streams): the committed rocprofv3 --stats summary of
        # this command, when the round has one; `live_over_rocprof` = this run's eager event time / that summary's kernel time
        rp = rocprof_families(f'bench_{args.model}' + ('' if args.size == 256 else f'_{args.size}'))
        if rp is not None:
            for f, v in rp['families'].items():
                if f in agg:
                    v['live_over_rocprof'] = (agg[f][1] / args.steps * 1e3) / v['kernel_ms_per_pass']
                    nb, fl = agg[f][3] / args.steps, agg[f][0] / args.steps
                    v['GB/s'], v['TFLOP/s'] = nb / v['kernel_ms_per_pass'] / 1e6, fl / v['kernel_ms_per_pass'] / 1e9
            res['roofline']['rocprof_timed_schedule'] = rp
        # step-level HBM traffic of the committed PMC passes (every kernel of one step) for step_roofline below
        if tsource:
            per_pass = json.load(open(os.path.join(REPO, tsource))).get('step_total_bytes')
            if per_pass:
                res['roofline']['step_traffic'] = {'hbm_bytes_per_step': per_pass, 'source': tsource}
    work = WORK_PER_PAIR.get((depth, args.size))
    if rank == 0 and work:
        # step level: algorithmic FLOP (3 x forward) and ideal-fusion bytes per frame-pair
        pps = pairs_per_step / world / (dt / args.steps)       # per GPU
        res['step_roofline'] = {'flop_per_pair': work[0], 'ideal_bytes_per_pair': work[1], 'TFLOP/s_per_gpu': pps * work[0] / 1e12,
                                'frac_of_mfma_peak': pps * work[0] / 1e12 / PEAK_BF16_TFLOPS, 'GB/s_per_gpu': pps * work[1] / 1e9,
                                'frac_of_hbm_peak': pps * work[1] / 1e9 / PEAK_HBM_GBS}
        st = res.get('roofline', {}).get('step_traffic')
        if st:      # PMC bytes of one step / ideal-fusion bytes of one step, and the rate those bytes moved at in the timed step
            ideal = work[1] * pairs_per_step / world
            res['step_roofline'].update({'traffic_bytes_per_step': st['hbm_bytes_per_step'], 'ideal_bytes_per_step': ideal,
                                         'traffic_ratio': st['hbm_bytes_per_step'] / ideal,
                                         'moved_TB/s': st['hbm_bytes_per_step'] / (dt / args.steps) / 1e12})
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('timing the CPU oracle (bounded sample) ...')
        res['cpu_baseline'] = cpu_baseline_subprocess(depth, args.size, min(os.cpu_count() or 1, 64))
    if world == 1 and not args.no_davis:
        # the second headline metric (BASELINE.json: "DAVIS J&F-Mean"; J&F itself needs the checkpoint + dataset, neither is on
        # the box): the label-propagation workload of configs[3] on the SAME model family, same process, after the train leg
        # - its own metric / value / roofline / cpu_baseline under the key "davis" of the same JSON line
        log(f'DAVIS leg: R{depth} {args.precision}, {args.davis_frames} propagated frames ...')
        res['davis'] = bench_davis(args, depth, dev, world, rank, steps=args.davis_frames, warmup=2)
    if world == 1 and rank == 0 and not args.no_extra_legs and not args.no_davis and args.model == 'r50' and args.size == 256:
        # BASELINE.json configs[1] and configs[4] beside the headline: after the timed region, each in its own
        # process, a few seconds of GPU time each
        del model, opt, imgs, batch
        eng.bufs.clear()
        torch.cuda.empty_cache()
        log('extra leg: ResNet-18 r2_1xNx8 config (T = 4), 256^2 ...')
        res['r18'] = extra_train_leg('r18', 256, args.batch, 10, 3)
        log('extra leg: ResNet-50 at 512^2 ...')
        res['r50_512'] = extra_train_leg('r50', 512, args.batch, 6, 2)
    if rank == 0:
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
