#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (/root/reference) on CPU.

Runs only in the build container (the reference does not travel to the GPU box).
The reference cannot be imported as a package there (``mmcv`` is not installed,
``mmaction/version.py`` is generated at install time), so this script registers a
minimal in-memory ``mmcv`` stand-in -- Registry/build_from_cfg, ``ConvModule`` =
conv(bias=False) -> norm -> act built from torch.nn layers, init helpers, PIL
``imresize`` -- and imports ONLY the hot-path modules of the reference, which then
execute their own code on torch-CPU.  Every number stored below is produced by the
reference's own functions (``ResNet``, ``SimSiamHead``, ``CosineSimLoss``,
``SimSiamBaseTracker``, ``VanillaTracker``, ``masked_attention_efficient``,
``spatial_neighbor``, ``pil_nearest_interpolate``) and torch.

Inputs and weights come from the closed-form fillers in oracle/vfs_oracle.py
(keyed by state_dict NAME), so the fixtures store outputs only.

Usage:  python tests/golden/gen_golden.py   (writes tests/golden/*.npz)
"""
import importlib
import os
import runpy
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)
from oracle.vfs_oracle import fill_state_dict_, fill_tensor  # noqa: E402


# ------------------------------------------------------------------ mmcv stand-in
def install_mmcv_standin():
    from PIL import Image

    class Registry:
        def __init__(self, name):
            self.name, self._d = name, {}

        def register_module(self, name=None, force=False, module=None):
            def deco(cls):
                self._d[name or cls.__name__] = cls
                return cls
            return deco(module) if module is not None else deco

        def get(self, k):
            return self._d.get(k)

        def __contains__(self, k):
            return k in self._d

    def build_from_cfg(cfg, registry, default_args=None):
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        cls = registry.get(args.pop('type'))
        return cls(**args)

    def kaiming_init(m, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
        nn.init.kaiming_normal_(m.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        if getattr(m, 'bias', None) is not None:
            nn.init.constant_(m.bias, bias)

    def constant_init(m, val, bias=0):
        if getattr(m, 'weight', None) is not None:
            nn.init.constant_(m.weight, val)
        if getattr(m, 'bias', None) is not None:
            nn.init.constant_(m.bias, bias)

    def build_norm_layer(cfg, num_features, postfix=''):
        cfg = dict(cfg)
        t = cfg.pop('type')
        rg = cfg.pop('requires_grad', True)
        cfg.setdefault('eps', 1e-5)
        cls = {'BN': nn.BatchNorm2d, 'BN1d': nn.BatchNorm1d, 'BN2d': nn.BatchNorm2d,
               'BN3d': nn.BatchNorm3d, 'SyncBN': nn.SyncBatchNorm}[t]
        layer = cls(num_features, **cfg)
        for p in layer.parameters():
            p.requires_grad = rg
        return 'bn' + str(postfix), layer

    class ConvModule(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                     groups=1, bias='auto', conv_cfg=None, norm_cfg=None,
                     act_cfg=dict(type='ReLU'), inplace=True, **kw):
            super().__init__()
            self.with_norm = norm_cfg is not None
            self.with_activation = act_cfg is not None
            if bias == 'auto':
                bias = not self.with_norm
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                                  padding=padding, dilation=dilation, groups=groups, bias=bias)
            if self.with_norm:
                self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
                self.add_module(self.norm_name, norm)
            if self.with_activation:
                assert act_cfg['type'] == 'ReLU'
                self.activate = nn.ReLU(inplace=act_cfg.get('inplace', inplace))
            kaiming_init(self.conv)
            if self.with_norm:
                constant_init(self.norm, 1, bias=0)

        @property
        def norm(self):
            return getattr(self, self.norm_name)

        def forward(self, x):
            x = self.conv(x)
            if self.with_norm:
                x = self.norm(x)
            if self.with_activation:
                x = self.activate(x)
            return x

    def imresize(img, size, return_scale=False, interpolation='bilinear', out=None, backend=None):
        assert backend == 'pillow' and interpolation == 'nearest'
        pil = Image.fromarray(img)
        return np.array(pil.resize(size, Image.NEAREST))

    def identity_deco(*a, **k):
        def d(f):
            return f
        return d

    mmcv = types.ModuleType('mmcv')
    mmcv.__path__ = []
    mmcv.imresize = imresize
    mmcv.is_seq_of = lambda seq, t, seq_type=None: all(isinstance(s, t) for s in seq)
    mmcv.mkdir_or_exist = lambda d, mode=0o777: os.makedirs(d, exist_ok=True)
    mmcv.BaseStorageBackend = type('BaseStorageBackend', (), {})

    class FileClient:
        @staticmethod
        def register_backend(name, backend=None, force=False):
            return (lambda c: c) if backend is None else None
    mmcv.FileClient = FileClient
    mmcv.load = mmcv.dump = lambda *a, **k: None
    mmcv.ProgressBar = object
    utils = types.ModuleType('mmcv.utils')
    utils.Registry, utils.build_from_cfg = Registry, build_from_cfg
    utils._BatchNorm = nn.modules.batchnorm._BatchNorm
    utils.print_log = lambda *a, **k: None
    import logging
    utils.get_logger = lambda name, log_file=None, log_level=logging.INFO: logging.getLogger(name)
    utils.collect_env = lambda: {}
    cnn = types.ModuleType('mmcv.cnn')
    cnn.ConvModule, cnn.build_norm_layer = ConvModule, build_norm_layer
    cnn.kaiming_init, cnn.constant_init = kaiming_init, constant_init
    cnn.normal_init = lambda m, mean=0, std=1, bias=0: None
    cnn.build_plugin_layer = lambda *a, **k: None
    cnn.build_activation_layer = lambda cfg: nn.ReLU()
    cnn.build_conv_layer = lambda cfg, *a, **k: nn.Conv2d(*a, **k)
    cnn.CONV_LAYERS = Registry('conv')
    cnn.NORM_LAYERS = Registry('norm')
    runner = types.ModuleType('mmcv.runner')
    runner.auto_fp16 = identity_deco
    runner.force_fp32 = identity_deco
    runner._load_checkpoint = lambda f, map_location=None: torch.load(f, map_location='cpu')
    runner.load_checkpoint = lambda *a, **k: None
    mmcv.utils, mmcv.cnn, mmcv.runner = utils, cnn, runner
    sys.modules.update({'mmcv': mmcv, 'mmcv.utils': utils, 'mmcv.cnn': cnn, 'mmcv.runner': runner})


def import_reference_hot_path():
    install_mmcv_standin()
    sys.path.insert(0, REF)
    for name in ['mmaction', 'mmaction.models', 'mmaction.models.backbones', 'mmaction.models.heads',
                 'mmaction.models.losses']:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, *name.split('.'))]
        sys.modules[name] = m
    models = sys.modules['mmaction.models']
    models.builder = importlib.import_module('mmaction.models.builder')
    models.registry = importlib.import_module('mmaction.models.registry')
    resnet = importlib.import_module('mmaction.models.backbones.resnet')
    sys.modules['mmaction.models.backbones'].ResNet = resnet.ResNet
    importlib.import_module('mmaction.models.losses.sim_loss')
    importlib.import_module('mmaction.models.heads.sim_siam_head')
    trackers = importlib.import_module('mmaction.models.trackers')
    common = importlib.import_module('mmaction.models.common')
    return models.builder, trackers, common


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def grads_summary(model):
    out = {}
    for n, p in model.named_parameters():
        g = p.grad
        out['gnorm/' + n] = np.float64(g.double().norm().item()) if g is not None else np.float64(-1)
        if g is not None:
            out['gsample/' + n] = g.flatten()[:: max(1, g.numel() // 16)][:16].numpy().copy()
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    builder, trackers, common = import_reference_hot_path()
    from mmaction.models.losses.sim_loss import CosineSimLoss
    from mmaction.models.backbones.resnet import ResNet
    from mmaction.models.common.affinity_utils import spatial_neighbor
    from mmaction.models.common.local_attention import masked_attention_efficient
    from mmaction.models.common.utils import pil_nearest_interpolate
    OUT = os.environ.get('VFS_GOLDEN_OUT', HERE)      # regenerate into a scratch directory to add single fixtures
    save = lambda name, **kw: (np.savez_compressed(os.path.join(OUT, name + '.npz'), **kw),
                               print('wrote', name, {k: getattr(v, 'shape', None) for k, v in kw.items()}))

    # ---- backbone forwards (train mode, batch statistics) ----
    for depth in (18, 50):
        net = ResNet(depth=depth, pretrained=None, out_indices=(0, 1, 2, 3),
                     norm_cfg=dict(type='SyncBN', requires_grad=True), zero_init_residual=True)
        net.init_weights()
        fill_state_dict_(net, seed=depth)
        net.train()
        x = fill_tensor([2, 3, 64, 64], seed=7, scale=2.0)
        outs = net(x)
        sd = net.state_dict()
        save(f'resnet{depth}_fwd', out0=outs[0].detach().numpy(), out1=outs[1].detach().numpy(),
             out2=outs[2].detach().numpy(), out3=outs[3].detach().numpy(),
             stem_running_mean=sd['conv1.bn.running_mean'].numpy(),
             stem_running_var=sd['conv1.bn.running_var'].numpy(),
             keys=np.array(list(sd.keys())))

    # ---- cosine loss ----
    p, z = fill_tensor([6, 32], 1), fill_tensor([6, 32], 2)
    save('cosine_loss', loss=CosineSimLoss(negative=False)(p, z).numpy(),
         loss_neg=CosineSimLoss(negative=True)(p, z).numpy())

    # ---- full train step on both shipped configs ----
    for tag, cfgname, shape in (('r18', 'r18_nc_sgd_cos_100e_r2_1xNx8_k400.py', [2, 2, 3, 4, 64, 64]),
                                ('r50', 'r50_nc_sgd_cos_100e_r5_1xNx2_k400.py', [4, 2, 3, 1, 64, 64])):
        cfg = runpy.run_path(os.path.join(REF, 'configs', cfgname))
        model = builder.build_model(cfg['model'], train_cfg=cfg['train_cfg'],
                                    test_cfg=AttrDict(cfg['test_cfg']))
        keys_at_init = list(model.state_dict().keys())
        init_stats = {'init/' + k: np.array([v.float().mean().item(), v.float().std().item()])
                      for k, v in model.state_dict().items() if v.numel() > 1}
        fill_state_dict_(model, seed=3)
        model.train()
        imgs = fill_tensor(shape, seed=11, scale=2.0)
        out = model.train_step(dict(imgs=imgs, label=torch.zeros(shape[0], 1)), None)
        out['loss'].backward()
        losses = model(imgs, return_loss=True)
        res = {'loss': np.float64(out['loss'].item()), 'num_samples': np.int64(out['num_samples']),
               'keys': np.array(keys_at_init), 'iteration': model.iteration.numpy()}
        for k, v in out['log_vars'].items():
            res['log/' + k] = np.float64(v)
        for k, v in losses.items():
            res['lossvec/' + k] = v.detach().numpy()
        res.update(grads_summary(model))
        res.update(init_stats)
        # one SGD step with the config's optimizer settings (torch.optim.SGD)
        oc = dict(cfg['optimizer'])
        assert oc.pop('type') == 'SGD'
        opt = torch.optim.SGD(model.parameters(), **oc)
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        opt.step()
        for n, p2 in model.named_parameters():
            d = (p2.detach() - before[n]).flatten()
            res['delta/' + n] = d[:: max(1, d.numel() // 8)][:8].numpy().copy()
        save(f'{tag}_train', **res)

    # ---- label propagation pieces ----
    m = spatial_neighbor(1, 12, 16, neighbor_range=8, device='cpu', dtype=torch.float32, mode='circle')
    save('spatial_neighbor', mask=np.packbits(m.numpy()), shape=np.array(m.shape))
    q = fill_tensor([1, 16, 12, 16], 21)
    k = fill_tensor([1, 16, 5, 12, 16], 22)
    v = fill_tensor([1, 3, 5, 12, 16], 23).abs()
    o = masked_attention_efficient(q, k, v, m, temperature=0.07, topk=10, normalize=True, non_mask_len=0)
    o_nomask = masked_attention_efficient(q, k, v, None, temperature=0.07, topk=10, normalize=True)
    save('masked_attention', out=o.numpy(), out_nomask=o_nomask.numpy())

    lab = (fill_tensor([480, 854], 31).numpy() * 2.5 + 2.5).astype(np.uint8)
    r = pil_nearest_interpolate(torch.from_numpy(lab)[None, None], size=(60, 107))
    save('pil_nearest', out=r[0, 0].numpy().astype(np.uint8))

    # ---- VanillaTracker.forward_test, R18 test-time config, small clip ----
    cfg = runpy.run_path(os.path.join(REF, 'configs', 'r18_nc_sgd_cos_100e_r2_1xNx8_k400.py'))
    tc = AttrDict(cfg['test_cfg'])
    tc['neighbor_range'] = 8          # 12x16 feature map
    tc['precede_frames'] = 3          # exercise the sliding window + duplicated first frame
    bb = dict(cfg['model']['backbone'])
    bb['out_indices'], bb['strides'] = tc['out_indices'], tc['strides']   # tools/test.py:129-133
    model = builder.build_model(dict(type='VanillaTracker', backbone=bb), train_cfg=None, test_cfg=tc)
    fill_state_dict_(model, seed=5)
    model.eval()
    T, H, W = 6, 96, 128
    imgs = fill_tensor([1, 1, 3, T, H, W], 41, scale=2.0)
    yy, xx = np.mgrid[0:H, 0:W]
    seg = np.zeros((H, W), np.uint8)
    seg[(yy > 20) & (yy < 60) & (xx > 30) & (xx < 70)] = 1
    seg[(yy > 50) & (yy < 90) & (xx > 80) & (xx < 120)] = 2
    with torch.no_grad():
        res = model(imgs, return_loss=False, ref_seg_map=torch.from_numpy(seg)[None],
                    img_meta=[dict(original_shape=(H, W, 3))])
        feat = model.extract_feat_test(common.video2images(imgs.reshape((-1,) + imgs.shape[2:])))
    save('forward_test_r18', seg_preds=res[0].astype(np.uint8), ref_seg=seg,
         feat_checksum=np.array([feat.double().sum().item(), feat.double().abs().sum().item()]),
         feat_shape=np.array(feat.shape), feat_sample=feat.flatten()[::997].numpy().copy())

    # ---- the same clip with test_cfg.all_blocks=True (README.md:76): one label map per res4 block ----
    model.test_cfg['all_blocks'] = True
    with torch.no_grad():
        res = model(imgs, return_loss=False, ref_seg_map=torch.from_numpy(seg)[None],
                    img_meta=[dict(original_shape=(H, W, 3))])
    save('forward_test_r18_all_blocks', seg_preds=res[0].astype(np.uint8), ref_seg=seg)


if __name__ == '__main__':
    main()
