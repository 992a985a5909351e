"""SimSiamHead under the reference's registry name and constructor
(mmaction/models/heads/sim_siam_head.py:14-174): avg-pool -> projector
(Linear+BN[+ReLU]) x num_projection_fcs -> z; predictor Linear+BN+ReLU, Linear -> p.
Linear layers run on the same MFMA implicit-GEMM kernel as the convolutions (1x1 "images"),
BN1d on the shared BatchNorm kernels."""
import torch
import torch.nn as nn

from .builder import build_loss
from .engine import BF16, ConvUnit
from .registry import HEADS


def _norm1d(cfg, n):
    t = cfg.get('type', 'BN')
    if t not in ('BN', 'BN1d', 'SyncBN'):
        raise KeyError(f'unsupported norm type {t}')
    return nn.BatchNorm1d(n, eps=cfg.get('eps', 1e-5))


@HEADS.register_module()
class SimSiamHead(nn.Module):
    def __init__(self, in_channels, conv_mid_channels=2048, conv_out_channles=2048, num_convs=0, kernel_size=1,
                 conv_cfg=dict(type='Conv2d'), norm_cfg=dict(type='BN'), act_cfg=None, drop_layer_cfg=None,
                 order=('pool', 'drop'), num_projection_fcs=3, projection_mid_channels=2048,
                 projection_out_channels=2048, drop_projection_fc=False, num_predictor_fcs=2,
                 predictor_mid_channels=512, predictor_out_channels=2048, drop_predictor_fc=False, with_norm=True,
                 loss_feat=dict(type='CosineSimLoss', negative=False), spatial_type='avg'):
        super().__init__()
        if num_convs != 0 or drop_layer_cfg is not None or drop_projection_fc or drop_predictor_fc \
                or spatial_type != 'avg':
            raise NotImplementedError('HIP path covers the shipped configs: no convs/dropout, spatial_type avg')
        assert set(order) == {'pool', 'drop'}
        self.in_channels, self.norm_cfg, self.with_norm = in_channels, norm_cfg, with_norm
        self.loss_feat = build_loss(loss_feat)
        self.spatial_type, self.order = spatial_type, order
        last = in_channels
        proj, self._plan = [], []          # plan: (seq name, linear idx, bn idx or None, relu)
        for i in range(num_projection_fcs):
            is_last = i == num_projection_fcs - 1
            out = projection_out_channels if is_last else projection_mid_channels
            self._plan.append(('projection_fcs', len(proj), len(proj) + 1, not is_last))
            proj += [nn.Linear(last, out), _norm1d(norm_cfg, out)]
            if not is_last:
                proj.append(nn.ReLU())
            last = out
        self.projection_fcs = nn.Sequential(*proj) if proj else nn.Identity()
        self._n_proj = num_projection_fcs
        pred = []
        for i in range(num_predictor_fcs):
            is_last = i == num_predictor_fcs - 1
            out = predictor_out_channels if is_last else predictor_mid_channels
            if is_last:
                self._plan.append(('predictor_fcs', len(pred), None, False))
                pred.append(nn.Linear(last, out))
            else:
                self._plan.append(('predictor_fcs', len(pred), len(pred) + 1, True))
                pred += [nn.Linear(last, out), _norm1d(norm_cfg, out), nn.ReLU()]
            last = out
        self.predictor_fcs = nn.Sequential(*pred) if pred else nn.Identity()
        self.avg_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.units = None
        self._engine = None
        from .engine import flush_counters_hook
        self.register_state_dict_pre_hook(flush_counters_hook)

    def init_weights(self):
        pass  # the reference keeps torch's Linear defaults (sim_siam_head.py:127-129)

    def attach(self, engine, prefix='img_head'):
        if self._engine is engine and self.units is not None:
            return
        self.units = []
        for seq, li, bi, relu in self._plan:
            lin = getattr(self, seq)[li]
            bn = getattr(self, seq)[bi] if bi is not None else None
            u = ConvUnit(f'{prefix}.{seq}.{li}', lin.weight, lin.bias, bn, 1, 1, 0, 'linear')
            u.relu = relu
            self.units.append(engine.register(u))
        self._engine = engine

    # ------------------------------------------------------------------ HIP execution
    def forward_nhwc(self, eng, feat, N, h, w, C, G, train):
        """feat bf16 [N,h,w,C] (G views stacked) -> z, p bf16 [N,Cout] + ctx."""
        dev = feat.device
        s = eng.stream(dev)
        x = eng.buf('img_head.pooled', (N, C), BF16, dev)
        eng.lib.avgpool_fwd(feat, x, N, h * w, C, s)
        ctx = dict(N=N, G=G, h=h, w=w, C=C, ins=[], raws=[], acts=[])
        a = x
        z = None
        for ui, u in enumerate(self.units):
            tr = train and (u.bn.training if u.bn is not None else True)
            ctx['ins'].append(a)
            raw, _, _ = eng.conv_fwd(u, a.view(N, 1, 1, u.cin), N, 1, 1, G, tr, defer_fin=u.bn is not None)
            raw = raw.view(N, u.cout)
            ctx['raws'].append(raw)
            if u.bn is not None:
                a = eng.bn_act(u, raw, N, G, tr, u.relu)
            else:
                a = raw
            ctx['acts'].append(a)
            if ui == self._n_proj - 1:
                z = a
        return z, a, ctx

    def backward_nhwc(self, eng, ctx, dp, dz=None):
        """dp: gradient wrt p (bf16 [N,C]); z only feeds the predictor (it is detached in the loss)."""
        N, G = ctx['N'], ctx['G']
        dev = dp.device
        g = dp
        for ui in range(len(self.units) - 1, -1, -1):
            u = self.units[ui]
            if ui == self._n_proj - 1 and dz is not None:
                raise NotImplementedError('explicit z gradient')
            if u.bn is not None:
                dx, _ = eng.bn_bwd(u, g, None, ctx['raws'][ui], N, G, relu=u.relu)
            else:
                dx = g
            g = eng.conv_bwd(u, dx, ctx['ins'][ui], N, 1, 1, 1, 1, need_dgrad=True)
            g = g.view(N, u.cin)
        gfeat = eng.buf('img_head.gfeat', (N, ctx['h'], ctx['w'], ctx['C']), BF16, dev)
        eng.lib.avgpool_bwd(g, gfeat, N, ctx['h'] * ctx['w'], ctx['C'], eng.stream(dev))
        return gfeat

    # ------------------------------------------------------------------ reference-compatible API
    def forward(self, x):
        """x [N,C,h,w] fp32 -> (z, p) fp32 (sim_siam_head.py:143-163); inference entry point."""
        from .engine import shared_engine
        if x.requires_grad:
            raise RuntimeError('SimSiamHead.forward is the inference entry point; training runs in the tracker')
        eng = shared_engine()
        self.attach(eng)
        eng.pack_weights()
        N, C, h, w = x.shape
        feat = x.permute(0, 2, 3, 1).contiguous().to(BF16)
        z, p, _ = self.forward_nhwc(eng, feat, N, h, w, C, 1, self.training)
        return z.float(), p.float()

    def loss(self, p1, z1, p2, z2, mask12=None, mask21=None, weight=1.):
        assert mask12 is None and mask21 is None
        loss_feat = self.loss_feat(p1, z2.detach()) * 0.5 + self.loss_feat(p2, z1.detach()) * 0.5
        return dict(loss_feat=loss_feat * weight)
