"""Golden vectors for BASELINE.json configs[0] AT ITS REAL SIZE (SURVEY.md section 8(d) "Cfg 1": the r18 config's model, 2-frame
SimSiam forward_train on 8 x 3 x 224 x 224 frames = imgs [4,2,3,1,224,224], CPU PyTorch), captured from the REAL reference in the
build container (same import recipe as gen_golden.py).  224 x 224 is the crop the shipped configs train on (configs/*:62): the
56 / 28 / 14 / 7-pixel feature maps are the ragged-tile cases of the HIP kernels.

    python tests/golden/gen_cfg1_golden.py          -> tests/golden/r18_cfg1_224.npz, tests/golden/r50_cfg_224.npz

Stored per model: loss vector, log_vars, per-parameter gradient norms + samples, one SGD step's deltas (as *_train.npz), plus
strided samples and checksums of the backbone's layer4 output for both views (train mode, batch statistics)."""
import os
import runpy
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G      # noqa: E402  (the mmcv stand-in + reference import recipe)
from gen_golden import fill_state_dict_, fill_tensor      # noqa: E402


def damp_block_outputs_(model):
    """the filler puts every BatchNorm scale around 1, the last one of each residual block included (the reference zero-initialises
    those): 8 / 16 undamped residual additions make the net ill-conditioned, and a bf16-storage implementation is then compared
    with the fp32 golden through amplified rounding noise.  x 0.25 on those scales (as tests/test_emu_train_step.py::_filled)."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            parts = name.split('.')
            if len(parts) >= 5 and parts[0] == 'backbone' and parts[1].startswith('layer') and parts[-2:] == ['bn', 'weight']:
                nconv = 2 if _is_basic(model) else 3
                if parts[3] == f'conv{nconv}':
                    p.mul_(0.25)


def _is_basic(model):
    return not hasattr(model.backbone.layer1[0], 'conv3')


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    builder, trackers, common = G.import_reference_hot_path()
    OUT = os.environ.get('VFS_GOLDEN_OUT', HERE)
    for name, cfgname, shape in (('r18_cfg1_224', 'r18_nc_sgd_cos_100e_r2_1xNx8_k400.py', [4, 2, 3, 1, 224, 224]),
                                 ('r50_cfg_224', 'r50_nc_sgd_cos_100e_r5_1xNx2_k400.py', [4, 2, 3, 1, 224, 224])):
        cfg = runpy.run_path(os.path.join(G.REF, 'configs', cfgname))
        model = builder.build_model(cfg['model'], train_cfg=cfg['train_cfg'], test_cfg=G.AttrDict(cfg['test_cfg']))
        fill_state_dict_(model, seed=3)
        damp_block_outputs_(model)
        model.train()
        imgs = fill_tensor(shape, seed=11, scale=2.0)
        feats = []
        hook = model.backbone.register_forward_hook(lambda m, i, o: feats.append(o.detach().clone()))
        out = model.train_step(dict(imgs=imgs, label=torch.zeros(shape[0], 1)), None)
        hook.remove()
        out['loss'].backward()
        res = {'loss': np.float64(out['loss'].item()), 'num_samples': np.int64(out['num_samples']), 'shape': np.array(shape)}
        for k, v in out['log_vars'].items():
            res['log/' + k] = np.float64(v)
        g_before = G.grads_summary(model)
        losses = model(imgs, return_loss=True)       # second forward: loss vectors (also moves the running statistics again)
        for k, v in losses.items():
            res['lossvec/' + k] = v.detach().numpy()
        res.update(g_before)
        assert len(feats) == 2      # one backbone call per view (sim_siam_base_tracker.py:66-69)
        for v, f in enumerate(feats):
            res[f'feat{v}/shape'] = np.array(f.shape)
            res[f'feat{v}/sample'] = f.flatten()[::37].numpy().copy()
            res[f'feat{v}/checksum'] = np.array([f.double().sum().item(), f.double().abs().sum().item()])
        oc = dict(cfg['optimizer'])
        assert oc.pop('type') == 'SGD'
        opt = torch.optim.SGD(model.parameters(), **oc)
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        opt.step()
        for n, p2 in model.named_parameters():
            d = (p2.detach() - before[n]).flatten()
            res['delta/' + n] = d[:: max(1, d.numel() // 8)][:8].numpy().copy()
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **res)
        print('wrote', name, 'loss', res['loss'])


if __name__ == '__main__':
    main()
