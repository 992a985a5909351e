"""ResNet backbone behind the reference's registry name and constructor
(mmaction/models/backbones/resnet.py:309-654), executing on the gfx950 HIP kernels.

Parameters live in torch.nn containers with the reference's module names so state_dicts are
interchangeable (`conv1.conv.weight`, `layer2.0.downsample.bn.running_mean`, ...); the arithmetic
is done by vfs_amd.engine on NHWC bf16 buffers.  There is no torch fallback."""
import os

import torch
import torch.nn as nn

from .engine import BF16, ConvUnit, Engine
from .registry import BACKBONES


def _kaiming_fan_out_relu_(w):
    """mmcv kaiming_init(conv): normal, mode fan_out, nonlinearity relu (resnet.py:537-538)."""
    nn.init.kaiming_normal_(w, a=0, mode='fan_out', nonlinearity='relu')


class ConvBN(nn.Module):
    """Parameter container with mmcv ConvModule's sub-module names (conv, bn)."""

    def __init__(self, cin, cout, k, stride, pad, norm_cfg, relu, dilation=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dilation, bias=False)
        ntype = norm_cfg.get('type', 'BN')
        if ntype not in ('BN', 'BN2d', 'SyncBN'):
            raise KeyError(f'unsupported norm type {ntype}')
        self.bn = nn.BatchNorm2d(cout, eps=norm_cfg.get('eps', 1e-5))
        self.sync = ntype == 'SyncBN'
        for p in self.bn.parameters():
            p.requires_grad = norm_cfg.get('requires_grad', True)
        self.relu = relu
        _kaiming_fan_out_relu_(self.conv.weight)
        self.unit = None

    @property
    def norm(self):
        return self.bn

    def forward(self, x):
        raise RuntimeError('ConvBN is a parameter container; run it through vfs_amd.engine')


class _Block(nn.Module):
    def __init__(self, convs, downsample):
        super().__init__()
        for i, c in enumerate(convs):
            self.add_module(f'conv{i + 1}', c)
        self.nconv = len(convs)
        self.downsample = downsample

    @property
    def convs(self):
        return [getattr(self, f'conv{i + 1}') for i in range(self.nconv)]


class BasicBlock(_Block):
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, norm_cfg, dilation=1, style='pytorch'):      # (style: Bottleneck only, resnet.py:15-73)
        super().__init__([ConvBN(inplanes, planes, 3, stride, dilation, norm_cfg, True, dilation),   # resnet.py:51-58
                          ConvBN(planes, planes, 3, 1, 1, norm_cfg, False)], downsample)


class Bottleneck(_Block):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, norm_cfg, dilation=1, style='pytorch'):
        # resnet.py:156-161: 'pytorch' puts the stride on the 3x3 conv, 'caffe' on the first 1x1 conv
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        super().__init__([ConvBN(inplanes, planes, 1, s1, 0, norm_cfg, True),
                          ConvBN(planes, planes, 3, s2, dilation, norm_cfg, True, dilation),     # resnet.py:172-179
                          ConvBN(planes, planes * 4, 1, 1, 0, norm_cfg, False)], downsample)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)),
                     50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)),
                     152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, pretrained=None, torchvision_pretrain=True, in_channels=3, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(3,), style='pytorch',
                 frozen_stages=-1, conv_cfg=dict(type='Conv'), norm_cfg=dict(type='BN2d', requires_grad=True),
                 act_cfg=dict(type='ReLU', inplace=True), norm_eval=False, partial_bn=False, with_cp=False,
                 zero_init_residual=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        assert 1 <= num_stages <= 4
        assert len(strides) == len(dilations) == num_stages
        assert max(out_indices) < num_stages
        assert style in ('pytorch', 'caffe')                                                  # resnet.py:154
        if in_channels != 3:
            raise NotImplementedError('HIP path covers in_channels=3 (the stem kernels read NHWC4 frames)')
        self.depth, self.pretrained, self.torchvision_pretrain = depth, pretrained, torchvision_pretrain
        self.in_channels, self.num_stages = in_channels, num_stages
        self.strides, self.dilations = tuple(strides), tuple(dilations)
        self.out_indices = tuple(out_indices)
        self.original_out_indices = tuple(out_indices)
        self.style, self.frozen_stages = style, frozen_stages
        self.norm_cfg, self.norm_eval, self.partial_bn = norm_cfg, norm_eval, partial_bn
        self.zero_init_residual = zero_init_residual
        self.block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]
        self.conv1 = ConvBN(3, 64, 7, 2, 3, norm_cfg, True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)   # parameter-free marker
        inplanes = 64
        self.res_layers = []
        for i, nb in enumerate(self.stage_blocks):
            planes, stride = 64 * 2 ** i, strides[i]
            down = None
            if stride != 1 or inplanes != planes * self.block.expansion:
                down = ConvBN(inplanes, planes * self.block.expansion, 1, stride, 0, norm_cfg, False)
            dil = dilations[i]          # make_res_layer (resnet.py:279-300): the first block of a dilated stage gets dil // 2
            blocks = [self.block(inplanes, planes, stride, down, norm_cfg, dil if dil == 1 else dil // 2, style=style)]
            inplanes = planes * self.block.expansion
            blocks += [self.block(inplanes, planes, 1, None, norm_cfg, dil, style=style) for _ in range(1, nb)]
            self.add_module(f'layer{i + 1}', nn.Sequential(*blocks))
            self.res_layers.append(f'layer{i + 1}')
        self.feat_dim = self.block.expansion * 64 * 2 ** (len(self.stage_blocks) - 1)
        self._engine = None
        self._freeze_stages()
        from .engine import flush_counters_hook
        self.register_state_dict_pre_hook(flush_counters_hook)

    # ------------------------------------------------------------------ reference-compatible API
    def init_weights(self):
        """resnet.py:525-553: checkpoint, or kaiming + BN(1,0) + zero-init of each block's last BN."""
        if isinstance(self.pretrained, str):
            ckpt = torch.load(self.pretrained, map_location='cpu')
            if self.torchvision_pretrain:
                self.load_torchvision_checkpoint(ckpt)
            else:
                # "ours" (resnet.py:536-539): mmcv load_checkpoint(self, path, strict=False) - the module's own key
                # names, an optional 'state_dict' wrapper and a 'module.' prefix left by (MM)DistributedDataParallel
                sd = ckpt.get('state_dict', ckpt)
                sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
                self.load_state_dict(sd, strict=False)
        elif self.pretrained is None:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    _kaiming_fan_out_relu_(m.weight)
                elif isinstance(m, nn.BatchNorm2d):
                    nn.init.constant_(m.weight, 1)
                    nn.init.constant_(m.bias, 0)
            if self.zero_init_residual:
                for m in self.modules():
                    if isinstance(m, _Block):
                        nn.init.constant_(m.convs[-1].bn.weight, 0)
        else:
            raise TypeError('pretrained must be a str or None')

    def load_torchvision_checkpoint(self, sd):
        """resnet.py:488-523 key mapping: layerX.Y.convN.{conv,bn} <- layerX.Y.{convN,bnN},
        downsample.{conv,bn} <- downsample.{0,1}."""
        if 'state_dict' in sd:
            sd = sd['state_dict']
        own = self.state_dict()
        for name, mod in self.named_modules():
            if not isinstance(mod, ConvBN):
                continue
            cname, bname = (name + '.0', name + '.1') if 'downsample' in name else (name, name.replace('conv', 'bn'))
            own[name + '.conv.weight'].copy_(sd[cname + '.weight'])
            for k in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
                if f'{bname}.{k}' in sd:
                    own[f'{name}.bn.{k}'].copy_(sd[f'{bname}.{k}'])

    @property
    def output_stride(self):
        s = 4
        for v in self.strides[:self.num_stages]:
            s *= v
        return s

    def switch_strides(self, strides=None):
        """resnet.py:624-637: change the stride of every stage's first block (+ its shortcut)."""
        for i, name in enumerate(self.res_layers):
            stride = self.strides[i] if strides is None else strides[i]
            blk = getattr(self, name)[0]
            if blk.downsample is None:
                continue
            tgt = blk.conv1 if (self.depth in (18, 34) or self.style == 'caffe') else blk.conv2      # resnet.py:624-637
            for m in (blk.downsample, tgt):
                m.conv.stride = (stride, stride)
                m.unit = None
        self._engine = None

    def switch_out_indices(self, out_indices=None):
        self.out_indices = self.original_out_indices if out_indices is None else tuple(out_indices)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.conv1.eval()
            for p in self.conv1.parameters():
                p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        if mode and self.partial_bn:
            count = 0
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    count += 1
                    if count >= 2:
                        m.eval()
                        m.weight.requires_grad = False
                        m.bias.requires_grad = False
        return self

    # ------------------------------------------------------------------ HIP execution
    def conv_modules(self):
        mods = [('conv1', self.conv1)]
        for lname in self.res_layers:
            for bi, blk in enumerate(getattr(self, lname)):
                for ci, c in enumerate(blk.convs):
                    mods.append((f'{lname}.{bi}.conv{ci + 1}', c))
                if blk.downsample is not None:
                    mods.append((f'{lname}.{bi}.downsample', blk.downsample))
        return mods

    def attach(self, engine, prefix='backbone'):
        """create the host-side unit of every layer on `engine` (idempotent)."""
        if self._engine is engine and all(m.unit is not None for _, m in self.conv_modules()):
            return
        for name, m in self.conv_modules():
            k = m.conv.kernel_size[0]
            u = ConvUnit(f'{prefix}.{name}', m.conv.weight, None, m.bn, k, m.conv.stride[0], m.conv.padding[0],
                         'stem' if name == 'conv1' else 'conv', dil=m.conv.dilation[0])
            u.need_wd = name != 'conv1'
            m.unit = engine.register(u)
        self._engine = engine

    def last_stage(self):
        return max(self.out_indices)

    def forward_nhwc(self, eng, x4, N, H, W_true, G, train, stop_after_out=True):
        """x4: NHWC4 bf16 [N,H,Wp,4].  Returns ({stage: (act, h, w, C)}, ctx for backward)."""
        ctx = dict(N=N, G=G, x4=x4, H=H, Wp=x4.shape[2], blocks=[])
        stem = self.conv1.unit
        stem.true_w = W_true
        stem_train = train and self.conv1.bn.training
        raw, Hs, Ws = eng.conv_fwd(stem, x4, N, H, x4.shape[2], G, stem_train)
        Hp, Wp = (Hs + 2 - 3) // 2 + 1, (Ws + 2 - 3) // 2 + 1
        dev = x4.device
        pooled = eng.buf('backbone.pool', (N, Hp, Wp, 64), BF16, dev)
        idx = eng.buf('backbone.pool_idx', (N, Hp, Wp, 64), torch.uint8, dev) if train else None
        # raw conv output at each argmax: the stem's BatchNorm backward reads it instead of gathering from raw
        xpool = eng.buf('backbone.pool_x', (N, Hp, Wp, 64), BF16, dev) if train else None
        npg = N // G if stem_train else N
        eng.timed('bn_relu_maxpool', (0.0, 2.0 * N * Hs * Ws * 64 + N * Hp * Wp * 64 * (2.0 + (3.0 if train else 0.0))), dev,
                  eng.lib.bn_relu_maxpool, raw, stem.bnp, pooled, idx, xpool, N, Hs, Ws, 64, Hp, Wp, npg, eng.stream(dev))
        ctx.update(stem_raw=raw, Hs=Hs, Ws=Ws, pooled=pooled, idx=idx, xpool=xpool, Hp=Hp, Wp2=Wp)
        x, h, w = pooled, Hp, Wp
        outs = {}
        for si, lname in enumerate(self.res_layers):
            for blk in getattr(self, lname):
                x, h, w, bctx = self._block_fwd(eng, blk, x, N, h, w, G, train)
                ctx['blocks'].append(bctx)
            if si in self.out_indices:
                outs[si] = (x, h, w, x.shape[-1])
            if stop_after_out and si >= self.last_stage():
                break   # the reference computes the remaining stages and discards them
        return outs, ctx

    def _block_fwd(self, eng, blk, x, N, h, w, G, train):
        convs = blk.convs
        bctx = dict(blk=blk, x=x, h=h, w=w, acts=[], raws=[], dims=[])
        a, ah, aw = x, h, w
        in_bn = None
        bctx['act_bn'] = []
        for ci, c in enumerate(convs):
            tr = train and c.bn.training
            oh, ow = c.unit.out_hw(ah, aw)
            last = ci == len(convs) - 1
            fold = (not last) and eng.can_fold_input_bn(convs[ci + 1].unit, N, G, oh, ow, train and convs[ci + 1].bn.training)
            # the statistics can be finished by the bn_act that follows directly (not by a folding consumer, and not
            # when the downsample conv - which re-uses the statistics workspace - runs in between)
            defer = (not fold) and not (last and blk.downsample is not None)
            raw, oh, ow = eng.conv_fwd(c.unit, a, N, ah, aw, G, tr, in_bn=in_bn, defer_fin=defer)
            bctx['raws'].append(raw)
            bctx['dims'].append((ah, aw, oh, ow))
            M = N * oh * ow
            in_bn = None
            if not last:
                nxt = convs[ci + 1]
                if fold:
                    # plain conv-BN-ReLU unit feeding a halo-tile conv: the activation is never written; the
                    # consumer (and its weight gradient) reads raw and applies scale/shift/ReLU while staging
                    in_bn = (c.unit.bnp, (N // G) if tr else N)
                    a = raw
                    bctx['acts'].append(raw)
                else:
                    a = eng.bn_act(c.unit, raw, M, G, tr, True)
                    bctx['acts'].append(a)
                bctx['act_bn'].append(in_bn)
                ah, aw = oh, ow
            else:
                if blk.downsample is not None:
                    d = blk.downsample
                    dtr = train and d.bn.training
                    draw, dh, dw = eng.conv_fwd(d.unit, x, N, h, w, G, dtr)
                    assert (dh, dw) == (oh, ow)
                    bctx['draw'] = draw
                    a = eng.bn_act(c.unit, raw, M, G, tr, True, rres=draw, rbnp=d.unit.bnp, want_mask=train)
                else:
                    a = eng.bn_act(c.unit, raw, M, G, tr, True, res=x, want_mask=train)
                bctx['out'] = a
                # what the backward needs of the block output: its ReLU mask (bit-packed when the engine wrote one)
                bctx['mask'] = c.unit.mask_bits if c.unit.mask_bits is not None else a
                ah, aw = oh, ow
        return a, ah, aw, bctx

    def backward_nhwc(self, eng, ctx, grads, on_stage_done=None):
        """grads: {stage: gradient wrt that stage's output (bf16 NHWC)}; accumulates parameter
        gradients (the stem needs no input gradient).  on_stage_done(module) is called as soon as
        all parameter gradients of a stage / the stem are final (gradient all-reduce overlap)."""
        N, G = ctx['N'], ctx['G']
        blocks = ctx['blocks']
        stage_end, stage_start = {}, {}
        bi = 0
        for si, lname in enumerate(self.res_layers):
            stage_start[bi] = si
            bi += len(getattr(self, lname))
            stage_end[bi - 1] = si
        # frozen_stages (resnet.py:577-599) freezes a PREFIX of the net: gradients are only propagated down to the first layer
        # that still has something to train
        def trainable(mod):
            return any(p.requires_grad for p in mod.parameters())
        stem_trains = trainable(self.conv1)
        first = 0 if stem_trains else next((i for i, b in enumerate(blocks) if trainable(b['blk'])), len(blocks))
        g = None
        notified = set()
        for i in range(len(blocks) - 1, -1, -1):
            if i < first:
                # a frozen prefix that ends INSIDE a stage: the stage's trainable blocks are final now (data parallel: their
                # gradients must still be all-reduced)
                si = max(s for b, s in stage_start.items() if b <= min(first, len(blocks) - 1))
                if on_stage_done is not None and si not in notified:
                    on_stage_done(getattr(self, self.res_layers[si]))
                break
            si = stage_end.get(i)
            if si is not None and si in grads:
                gs = grads[si]
                if g is None:
                    g = gs
                else:
                    raise NotImplementedError('gradients at several stages')  # not on the VFS path
            if g is None:
                continue
            # the BatchNorm unit that consumes this block's INPUT gradient: the join unit of the block before
            prev = blocks[i - 1] if i > 0 else None
            next_bn = None if prev is None else (prev['blk'].convs[-1].unit, prev['raws'][-1], prev['mask'])
            g = self._block_bwd(eng, blocks[i], g, N, G, next_bn, need_input_grad=stem_trains or i > first)
            if on_stage_done is not None and i in stage_start:
                notified.add(stage_start[i])
                on_stage_done(getattr(self, self.res_layers[stage_start[i]]))
        if not stem_trains:
            if on_stage_done is not None:
                on_stage_done(self.conv1)
            return
        # stem: maxpool+relu backward -> BN backward -> wgrad
        dev = g.device
        Hs, Ws = ctx['Hs'], ctx['Ws']
        stem = self.conv1.unit
        count = eng.stem_pool_bn_bwd(stem, g, ctx['pooled'], ctx['idx'], ctx['stem_raw'], N, Hs, Ws, ctx['Hp'], ctx['Wp2'], G,
                                     xpool=ctx.get('xpool'))
        if os.environ.get('VFS_STEM_FUSED', '1') == '1':
            eng.stem_wgrad_fused(stem, ctx['x4'], ctx['H'], ctx['Wp'], g, ctx['pooled'], ctx['idx'], ctx['stem_raw'],
                                 N, Hs, Ws, ctx['Hp'], ctx['Wp2'], G, count)
        else:   # materialise dx, then the generic implicit-GEMM stem wgrad
            dx = eng.buf('backbone.conv1.dx', ctx['stem_raw'].shape, BF16, dev)
            eng.lib.stem_pool_bn_bwd_apply(g, ctx['pooled'], ctx['idx'], ctx['stem_raw'], stem.bnp, stem.bsums, dx, N, Hs,
                                           Ws, 64, ctx['Hp'], ctx['Wp2'], (N // G) if stem.bn.training else N, count, eng.stream(dev))
            eng.conv_bwd(stem, dx, ctx['x4'], N, ctx['H'], ctx['Wp'], Hs, Ws, need_dgrad=False)
        if on_stage_done is not None:
            on_stage_done(self.conv1)

    def _block_bwd(self, eng, bctx, g, N, G, next_bn=None, need_input_grad=True):
        """next_bn = (unit, raw, out) of the preceding block's join: the dgrad that completes this block's
        input gradient also emits that unit's BatchNorm-backward statistics (Engine.conv_bwd).
        need_input_grad=False (everything below is frozen): the dgrads towards the block input are skipped."""
        def groups(unit):      # statistics groups of a unit's BatchNorm backward: one when it runs in eval mode
            return G if unit.bn.training else 1
        blk = bctx['blk']
        convs = blk.convs
        last = len(convs) - 1
        h, w = bctx['h'], bctx['w']
        ih, iw, oh, ow = bctx['dims'][last]
        M = N * oh * ow
        # join: y = relu(bn_last(raw) + identity)
        dx, gm = eng.bn_bwd(convs[last].unit, g, bctx['mask'], bctx['raws'][last], M, G, want_gm=True)
        gmask = None
        if gm is None:       # bit-packed mask: the masked gradient g * (y > 0) is applied on the fly by its consumers
            gm, gmask = g, bctx['mask']
        if blk.downsample is not None:
            ddx, _ = eng.bn_bwd(blk.downsample.unit, gm, gmask, bctx['draw'], M, G)
        for ci in range(last, -1, -1):
            c = convs[ci]
            ih, iw, oh, ow = bctx['dims'][ci]
            x_in = bctx['x'] if ci == 0 else bctx['acts'][ci - 1]
            add = gm if (ci == 0 and blk.downsample is None) else None
            if ci > 0:       # plain conv-BN-ReLU unit in front: its statistics come out of this dgrad
                bn_next = (convs[ci - 1].unit, bctx['raws'][ci - 1], None, True, groups(convs[ci - 1].unit))
            elif blk.downsample is None and next_bn is not None:
                bn_next = (next_bn[0], next_bn[1], next_bn[2], True, groups(next_bn[0]))
            else:
                bn_next = None
            x_in_bn = bctx['act_bn'][ci - 1] if ci > 0 else None
            gin = eng.conv_bwd(c.unit, dx, x_in, N, ih, iw, oh, ow, need_dgrad=(ci > 0 or need_input_grad), add=add, bn_next=bn_next,
                               x_in_bn=x_in_bn, add_mask=gmask if add is not None else None)
            if ci > 0:
                p = convs[ci - 1]
                _, _, ph, pw = bctx['dims'][ci - 1]
                dx, _ = eng.bn_bwd(p.unit, gin, None, bctx['raws'][ci - 1], N * ph * pw, G, relu=True)
        if blk.downsample is not None:
            d = blk.downsample
            boh, bow = bctx['dims'][last][2:]
            bn_next = None if next_bn is None else (next_bn[0], next_bn[1], next_bn[2], True, groups(next_bn[0]))
            gin = eng.conv_bwd(d.unit, ddx, bctx['x'], N, h, w, boh, bow, need_dgrad=need_input_grad, add=gin, g_out=gin, bn_next=bn_next)
        return gin

    # ------------------------------------------------------------------ module-level forward
    def forward(self, x):
        """x [N,3,H,W] fp32 -> stage output(s) [N,C,h,w] fp32 (resnet.py:555-575).  Inference /
        feature extraction entry point; training goes through the tracker's fused step."""
        from .engine import shared_engine
        if x.requires_grad:
            raise RuntimeError('ResNet.forward is the inference entry point; use SimSiamBaseTracker.forward_train')
        eng = shared_engine(x.device)
        N, _, H, W = x.shape
        precision = getattr(self, 'eval_precision', None) or os.environ.get('VFS_EVAL_PRECISION', 'fp32')
        if not self.training and precision == 'fp32':
            # eval mode: the reference's fp32 arithmetic, bit-defined (vfs_amd/exact.py, csrc/exact_f32.hip)
            from .exact import exact_state
            x4 = eng.buf('exact.x4', (N, H, W, 4), torch.float32, x.device)
            eng.lib.imgs_to_nhwc4_f32(x.contiguous().float(), x4, N, 1, 1, H, W, eng.stream(x.device))
            outs, _ = exact_state(self).forward(eng, x4, N, H, W, stop_after_out=True)
            res = [outs[i][0].permute(0, 3, 1, 2).contiguous() for i in sorted(outs)]
            return res[0] if len(res) == 1 else tuple(res)
        self.attach(eng)
        eng.pack_weights()
        Wp = W + (W & 1)
        x4 = eng.buf('backbone.x4', (N, H, Wp, 4), BF16, x.device)
        eng.lib.imgs_to_nhwc4(x.contiguous().float(), x4, N, 1, 1, H, W, Wp, eng.stream(x.device))
        train = self.training
        if train:
            from .engine import bump_params_epoch
            bump_params_epoch()      # running statistics change through raw pointers
        outs, _ = self.forward_nhwc(eng, x4, N, H, W, 1, train, stop_after_out=True)
        res = [outs[i][0].float().permute(0, 3, 1, 2).contiguous() for i in sorted(outs)]
        return res[0] if len(res) == 1 else tuple(res)
