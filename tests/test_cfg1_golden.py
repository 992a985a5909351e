"""BASELINE.json configs[0] at its real size (SURVEY 8(d) "Cfg 1": r18 model, imgs [4,2,3,1,224,224]) and the r50 model at the same
224 x 224 crop (what configs/r*_*.py:62 train on), against vectors captured from the REAL reference by
tests/golden/gen_cfg1_golden.py.  Two hops:
  * CPU (`-m "not gpu"`): the oracle is pinned to the golden at this size (fp32, 1e-5 class bars);
  * GPU (`-m gpu`): the HIP train step (bf16 storage) directly against the golden - loss, per-view layer4 features, every
    parameter-gradient norm - with the oracle's own bf16-storage emulation on the same input as the bar, as everywhere else."""
import os

import numpy as np
import pytest
import torch

from oracle import vfs_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [('r18_cfg1_224', 18), ('r50_cfg_224', 50)]


def _load(name):
    return np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz'), allow_pickle=False)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _filled(depth):
    """the weights of the golden: closed-form filler, the last BatchNorm scale of every residual block x 0.25
    (gen_cfg1_golden.py::damp_block_outputs_)"""
    ref = O.build_tracker(depth)
    O.fill_state_dict_(ref, seed=3)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, O.BasicBlock):
                m.conv2.bn.weight.mul_(0.25)
            elif isinstance(m, O.Bottleneck):
                m.conv3.bn.weight.mul_(0.25)
    return ref


def _oracle_step(depth, imgs, emulate, stats='stored'):
    ref = _filled(depth)
    ref.set_emulate_bf16(emulate, stats=stats).train()
    feats = []
    hook = ref.backbone.register_forward_hook(lambda m, i, o: feats.append(o.detach().clone()))
    losses = ref.forward_train(imgs)
    hook.remove()
    loss, log_vars = O.parse_losses(losses)
    loss.backward()
    return ref, losses, log_vars, feats


@pytest.mark.parametrize('name,depth', CASES)
def test_oracle_matches_reference_at_224(name, depth):
    g = _load(name)
    shape = [int(v) for v in g['shape']]
    imgs = O.fill_tensor(shape, seed=11, scale=2.0)
    ref, losses, log_vars, feats = _oracle_step(depth, imgs, False)
    assert abs(log_vars['loss'] - float(g['loss'])) < 1e-5 * max(1.0, abs(float(g['loss'])))
    for k, v in log_vars.items():
        assert abs(v - float(g['log/' + k])) < 1e-5, k
    assert len(feats) == 2
    for v, f in enumerate(feats):
        assert list(f.shape) == [int(s) for s in g[f'feat{v}/shape']]
        assert _rel(f.flatten()[::37].numpy(), g[f'feat{v}/sample']) < 2e-5, v
        assert abs(f.double().abs().sum().item() - g[f'feat{v}/checksum'][1]) < 1e-5 * g[f'feat{v}/checksum'][1]
    bad = []
    for n, p in ref.named_parameters():
        gn, mine = float(g['gnorm/' + n]), float(p.grad.double().norm())
        if abs(mine - gn) > 2e-3 * max(gn, 1e-6) + 1e-7:
            bad.append((n, mine, gn))
        s = p.grad.flatten()[:: max(1, p.grad.numel() // 16)][:16].numpy()
        want = g['gsample/' + n]
        assert np.abs(s - want).max() <= 2e-3 * max(np.abs(want).max(), 1e-6) + 1e-7, n
    assert not bad, bad[:5]
    # the loss VECTORS of the golden come from a second forward (running statistics moved once more; batch statistics and
    # therefore the values are the same): unreduced [B*T] rows under the reference's keys
    for k, v in losses.items():
        assert list(v.shape) == [shape[0] * shape[3]] and _rel(v.detach().numpy(), g['lossvec/' + k]) < 2e-5, k
    params = [p for _, p in ref.named_parameters()]
    before = [p.detach().clone() for p in params]
    with torch.no_grad():
        O.sgd_step(params, [p.grad for p in params], [None] * len(params), lr=0.05)
    for (n, p), b in zip(ref.named_parameters(), before):
        d = (p.detach() - b).flatten()
        d = d[:: max(1, d.numel() // 8)][:8].numpy()
        want = g['delta/' + n]
        assert np.abs(d - want).max() <= 2e-3 * max(np.abs(want).max(), 1e-9) + 1e-9, n


def _jitter(imgs, seed):
    """every frame value moved by one fp32 ulp up or down (seeded): the same scene, other bf16 rounding flips downstream"""
    s = torch.randint(0, 2, imgs.shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1
    return torch.nextafter(imgs, imgs + s.float() * imgs.abs().clamp_min(1e-30))


DRAWS = (('stored', 'stored', 0), ('engine', 'engine', 0), ('acc', 'acc', 0), ('jitter', 'stored', 1))


def _emulation_draws(depth, imgs, want):
    """The yardstick of the end-to-end comparisons: SEVERAL bf16-storage emulations of the step.  Every draw rounds the same
    tensors to bf16; they differ in where the BatchNorm statistics are taken (ConvBN.emulate_stats) or by one fp32 ulp of the
    input.  tools/parity_noise_draws.py (profiles/r04_parity_noise_draws_*.json): at 4 frames per view the median over the
    parameters of err_a / err_b between two such draws scatters from 0.45 to 2.2 - ONE draw (round 3's yardstick, against which
    the HIP step measured 1.44) cannot resolve anything below ~2x; at 32 frames per view the scatter is 0.75 .. 1.33.
    want(ref, log, feats) -> dict of error figures of one draw."""
    out = []
    for label, stats, jit in DRAWS:
        ref, _, log, feats = _oracle_step(depth, _jitter(imgs, jit) if jit else imgs, True, stats=stats)
        out.append((label, want(ref, log, feats)))
    return out


def _hip_step(gpu_backend, depth, imgs):
    import vfs_amd
    dev = gpu_backend.dev
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_r{depth}.py'))
    model = vfs_amd.build_model(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    model.load_state_dict(_filled(depth).state_dict())
    model.to(dev).train()
    out = model.train_step(dict(imgs=imgs.to(dev), label=torch.zeros(imgs.shape[0], 1)), None)
    out['loss'].backward()
    blocks = model.backbone.layer4
    feat = gpu_backend.eng.bufs[f'backbone.layer4.{len(blocks) - 1}.conv{blocks[0].nconv}.act'].float().cpu().permute(0, 3, 1, 2).contiguous()
    return model, out, feat


def _keep(name, table):
    try:
        import json
        os.makedirs(os.path.join(REPO, 'gpurun_out', 'parity'), exist_ok=True)
        json.dump(table, open(os.path.join(REPO, 'gpurun_out', 'parity', f'golden_{name}.json'), 'w'), indent=1, default=str)
    except OSError:
        pass


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


@pytest.mark.gpu
@pytest.mark.parametrize('name,depth', CASES)
def test_hip_train_step_vs_reference_golden_at_224(gpu_backend, name, depth):
    """The HIP step against vectors captured from the REAL reference (fp32), 4 (ResNet-18) / 8 (ResNet-50) frames per view.  End to
    end through 17 / 50 bf16-stored layers and BatchNorm1d batches of FOUR samples the error is amplified rounding noise; the
    yardstick is the SCATTER of four bf16-storage emulations of the same step (_emulation_draws), not one of them:
      * loss / layer4 features: HIP error <= 1.2 x the largest emulation error (features: 1.9e-2 / 4.8e-2 rel-L2 in every draw);
      * every parameter-gradient norm: HIP error <= 3 x the largest emulation error of that parameter + 0.15;
      * median over the parameters of (HIP error / rms of the emulations' errors) <= 2 x the largest value the same statistic
        takes for an emulation held out of the rms (leave-one-out) - "the HIP step is one more draw of the same noise".
    These bars are as loose as round 3's ON PURPOSE: in the build container further legitimate draws (other ulp jitters) reach
    2.1 for the median statistic against 0.5 .. 1.3 for the four draws used here, and exceed 1.2 x the largest draw by up to
    0.29 on single parameters - at four frames per view nothing tighter is honest.  The end-to-end bar that can resolve a
    defect is the 32-frames-per-view test below.  The kernels themselves are held to bf16 rounding, entry by entry, by the per-stage comparison at this very size
    (tests/test_emu_train_step.py::test_every_stage_matches_oracle_at_bench_sizes[*_224_*])."""
    g = _load(name)
    shape = [int(v) for v in g['shape']]
    imgs = O.fill_tensor(shape, seed=11, scale=2.0)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    gold_norm = {k[6:]: float(g[k]) for k in g.files if k.startswith('gnorm/')}
    Nv = shape[0] * shape[3]

    def errors(params, log, feats):
        e = dict(loss=abs(float(log['loss']) - float(g['loss'])))
        for v in range(2):
            e[f'feat{v}'] = _l2(feats[v].flatten()[::37].numpy(), g[f'feat{v}/sample'])
        e['norm'] = {n: abs(float(p.grad.double().norm()) - gold_norm[n]) / gold_norm[n] for n, p in params if gold_norm[n] >= 1e-6}
        return e
    draws = _emulation_draws(depth, imgs, lambda ref, log, feats: errors(ref.named_parameters(), log, feats))
    model, out, feat = _hip_step(gpu_backend, depth, imgs)
    assert out['num_samples'] == shape[0]
    assert list(out['log_vars'].keys()) == [k[4:] for k in g.files if k.startswith('log/')]
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        if gold_norm[n] < 1e-6:
            assert float(p.grad.norm()) < 5e-3, n
    mine = errors(list(model.named_parameters()), out['log_vars'], [feat[:Nv], feat[Nv:]])
    worst = {k: max(d[k] for _, d in draws) for k in ('loss', 'feat0', 'feat1')}
    table = {'loss error HIP / largest emulation draw': (mine['loss'], worst['loss'])}
    failed = []
    if not mine['loss'] <= 1.2 * worst['loss'] + 2e-3:
        failed.append(('loss', mine['loss'], worst['loss']))
    for v in range(2):
        k = f'feat{v}'
        table[f'layer4 features view {v}: rel-L2 HIP / emulation draws'] = (mine[k], [d[k] for _, d in draws])
        if not mine[k] <= 1.2 * worst[k]:
            failed.append((k, mine[k], worst[k]))
    names = list(mine['norm'])
    rms = {n: (sum(d['norm'][n] ** 2 for _, d in draws) / len(draws)) ** 0.5 for n in names}
    top = {n: max(d['norm'][n] for _, d in draws) for n in names}
    for n in names:
        if not mine['norm'][n] <= 3.0 * top[n] + 0.15:
            failed.append((n, mine['norm'][n], top[n]))
    z_hip = _median(mine['norm'][n] / max(rms[n], 1e-3) for n in names)
    z_loo = []
    for j, (label, d) in enumerate(draws):
        others = [dd for i, (_, dd) in enumerate(draws) if i != j]
        r = {n: (sum(o['norm'][n] ** 2 for o in others) / len(others)) ** 0.5 for n in names}
        z_loo.append((label, _median(d['norm'][n] / max(r[n], 1e-3) for n in names)))
    wn = max(names, key=lambda n: mine['norm'][n])
    table['parameter-gradient norms: worst relative error vs golden (HIP, largest emulation draw of that parameter)'] = (wn, mine['norm'][wn], top[wn])
    table['largest (HIP error - 3 x largest draw) over the parameters (bar 0.15)'] = max(mine['norm'][n] - 3.0 * top[n] for n in names)
    table['median over the parameters of HIP error / rms of the emulation draws'] = z_hip
    table['the same statistic for each draw held out (leave-one-out)'] = z_loo
    print(name, table)
    _keep(name, table)
    if not z_hip <= 2.0 * max(z for _, z in z_loo):
        failed.append(('median ratio', z_hip, z_loo))
    assert not failed, failed


@pytest.mark.gpu
@pytest.mark.parametrize('depth', [18, 50])
def test_hip_train_step_end_to_end_vs_fp32_oracle_32_frames_per_view(gpu_backend, depth):
    """End to end where the comparison CAN resolve something: ResNet-18 and (round 5) ResNet-50 - the headline configuration's
    network at its batch - imgs [32,2,3,1,224,224] (BatchNorm batches of 32 frames, the bench's batch).  Reference = the oracle in fp32 (pinned to the real reference at this crop by test_oracle_matches_reference_at_224),
    yardstick = four bf16-storage emulation draws (their error ratios scatter 0.75 .. 1.33 at this size, build container).
    HIP vs fp32: loss <= 3 x the largest draw + 2e-4, layer4 features <= 1.1 x the largest draw (1.89e-2 in every draw: what is
    left is bf16 storage), every gradient norm <= 1.5 x the largest draw + 0.05, median gradient-norm error <= 1.5 x the largest
    draw's median (draws: 0.7 - 0.9 %).  Calibration (build container, four further draws as stand-ins for the HIP step): medians
    0.5 - 0.8 %, single parameters up to 0.039 above 1.5 x the largest draw, features 1.0009 x, loss 0.9 x."""
    shape = [32, 2, 3, 1, 224, 224]
    imgs = O.fill_tensor(shape, seed=11, scale=2.0)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    ref, _, log32, feats32 = _oracle_step(depth, imgs, False)
    gnorm = {n: float(p.grad.double().norm()) for n, p in ref.named_parameters()}
    Nv = shape[0]

    def errors(params, log, feats):
        e = dict(loss=abs(float(log['loss']) - float(log32['loss'])))
        for v in range(2):
            e[f'feat{v}'] = float((feats[v].double() - feats32[v].double()).norm() / feats32[v].double().norm())
        e['norm'] = {n: abs(float(p.grad.double().norm()) - gnorm[n]) / gnorm[n] for n, p in params if gnorm[n] >= 1e-6}
        return e
    draws = _emulation_draws(depth, imgs, lambda r, log, feats: errors(r.named_parameters(), log, feats))
    model, out, feat = _hip_step(gpu_backend, depth, imgs)
    mine = errors(list(model.named_parameters()), out['log_vars'], [feat[:Nv], feat[Nv:]])
    table, failed = {}, []
    wl = max(d['loss'] for _, d in draws)
    table['loss error HIP / largest draw'] = (mine['loss'], wl)
    # ONE scalar per draw: four draws under-sample its scatter.  tools/parity_loss_scatter.py (round 5, advisor finding r04: the HIP
    # loss sat ~2x outside the four draws): TWELVE emulation draws of the ResNet-18 step (three statistics variants x one-ulp input
    # jitters) land 5.7e-5 .. 1.1e-3 from the fp32 loss, median 4.4e-4 (profiles/r05_parity_loss_scatter_r18.json) - the MI355X's
    # 9.4e-4 (ResNet-18) / 4.0e-4 (ResNet-50) is inside that range; the bar is 3 x the largest of the four draws run here + 2e-4
    if not mine['loss'] <= 3.0 * wl + 2e-4:
        failed.append(('loss', mine['loss'], wl))
    for v in range(2):
        k = f'feat{v}'
        w = max(d[k] for _, d in draws)
        table[f'layer4 features view {v}: rel-L2 HIP / draws'] = (mine[k], [d[k] for _, d in draws])
        if not mine[k] <= 1.1 * w:
            failed.append((k, mine[k], w))
    names = list(mine['norm'])
    top = {n: max(d['norm'][n] for _, d in draws) for n in names}
    for n in names:
        if not mine['norm'][n] <= 1.5 * top[n] + 0.05:
            failed.append((n, mine['norm'][n], top[n]))
    med_hip, med_draws = _median(mine['norm'].values()), [(label, _median(d['norm'].values())) for label, d in draws]
    table['median gradient-norm error HIP / draws'] = (med_hip, med_draws)
    table['worst gradient-norm error HIP'] = max((mine['norm'][n], n) for n in names)
    table['largest (HIP error - 1.5 x largest draw) over the parameters (bar 0.05)'] = max(mine['norm'][n] - 1.5 * top[n] for n in names)
    print(table)
    _keep(f'r{depth}_b32_224_end_to_end', table)
    if not med_hip <= 1.5 * max(m for _, m in med_draws):
        failed.append(('median', med_hip, med_draws))
    assert not failed, failed
