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


def _oracle_step(depth, imgs, emulate):
    ref = _filled(depth)
    ref.set_emulate_bf16(emulate).train()
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


@pytest.mark.gpu
@pytest.mark.parametrize('name,depth', CASES)
def test_hip_train_step_vs_reference_golden_at_224(gpu_backend, name, depth):
    import vfs_amd
    g = _load(name)
    shape = [int(v) for v in g['shape']]
    dev = gpu_backend.dev
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_r{depth}.py'))
    model = vfs_amd.build_model(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    model.load_state_dict(_filled(depth).state_dict())
    model.to(dev).train()
    imgs = O.fill_tensor(shape, seed=11, scale=2.0)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    refbf, _, logbf, featbf = _oracle_step(depth, imgs, True)      # what bf16 storage costs on this input

    out = model.train_step(dict(imgs=imgs.to(dev), label=torch.zeros(shape[0], 1)), None)
    out['loss'].backward()
    assert out['num_samples'] == shape[0]
    assert list(out['log_vars'].keys()) == [k[4:] for k in g.files if k.startswith('log/')]

    # End to end through 17 / 50 bf16-stored layers and a head whose BatchNorm1d batches hold FOUR samples: the error is amplified
    # rounding noise, and the oracle's bf16-storage emulation is ONE other draw of that noise - two draws differ by small factors
    # (measured on the MI355X: up to 2.5x on single entries).  Hence 3x + a floor here; the kernels themselves are held to
    # bf16 rounding by the per-stage comparison at this very size (test_every_stage_matches_oracle_at_bench_sizes[*_224_*]).
    def bar(mine, emu):
        return mine <= 3.0 * emu + 5e-3
    for k, v in out['log_vars'].items():
        want = float(g['log/' + k])
        assert bar(abs(v - want), abs(logbf[k] - want)), (k, v, want, logbf[k])
    # layer4 output of both views = the last block's join output in the engine's pool: [2*Nv, h, w, C] NHWC bf16 -> NCHW
    blocks = model.backbone.layer4
    feat = gpu_backend.eng.bufs[f'backbone.layer4.{len(blocks) - 1}.conv{blocks[0].nconv}.act'].float().cpu().permute(0, 3, 1, 2).contiguous()
    Nv = shape[0] * shape[3]
    table = {}
    for v in range(2):
        want = g[f'feat{v}/sample']
        mine = feat[v * Nv:(v + 1) * Nv].flatten()[::37].numpy()
        emu = featbf[v].flatten()[::37].numpy()
        table[f'layer4 features view {v}: rel-L2 HIP / oracle bf16 emulation'] = (_l2(mine, want), _l2(emu, want))
        assert bar(_l2(mine, want), _l2(emu, want)), (v, _l2(mine, want), _l2(emu, want))
    # Parameter gradients.  At this size (4 frames per view: the head's BatchNorm1d batches hold FOUR samples) single gradient ENTRIES
    # are not resolvable in bf16 storage at all: the oracle's own bf16-storage emulation misses the fp32 golden by 60 % (ResNet-18) /
    # 108 % (ResNet-50) of the largest entry in the MEDIAN over the parameters, and the gradient NORMS by 5 % / 11 % (measured in the
    # build container).  So only the norms are compared, against that emulation's error as the yardstick; the kernels themselves are
    # held to bf16 rounding, entry by entry, by the per-stage comparison at this very size
    # (tests/test_emu_train_step.py::test_every_stage_matches_oracle_at_bench_sizes[*_224_*]).
    gbf = dict(refbf.named_parameters())
    ratios, worst = [], (0.0, None)
    for n, p in model.named_parameters():
        gn = float(g['gnorm/' + n])
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        if gn < 1e-6:
            assert float(p.grad.norm()) < 5e-3, n
            continue
        mine = abs(float(p.grad.double().norm()) - gn) / gn
        emu = abs(float(gbf[n].grad.double().norm()) - gn) / gn
        assert mine <= 3.0 * emu + 0.15, (n, mine, emu)
        ratios.append(mine / max(emu, 1e-3))
        if mine > worst[0]:
            worst = (mine, n)
    table['parameter-gradient norms: worst relative error vs golden (HIP)'] = worst
    ratios.sort()
    table['median (HIP error / bf16-emulation error) over parameter-gradient norms'] = ratios[len(ratios) // 2]
    print(name, table)
    try:
        import json
        os.makedirs(os.path.join(REPO, 'gpurun_out', 'parity'), exist_ok=True)
        json.dump(table, open(os.path.join(REPO, 'gpurun_out', 'parity', f'golden_{name}.json'), 'w'), indent=1, default=str)
    except OSError:
        pass
    assert ratios[len(ratios) // 2] < 2.0
