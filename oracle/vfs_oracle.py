"""CPU oracle for the VFS hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``vfs_amd/`` imports it; the product
path fails loudly when ``libvfs_hip.so`` is missing instead of falling back here.

What it is: an independent restatement, in plain PyTorch-CPU fp32, of the
arithmetic the reference executes on the path BASELINE.json names.  The
reference is 100 % Python and delegates every number to third-party
``torch`` (pinned 1.6.0 in docker/Dockerfile:69; 2.10.0 CPU here) and
``mmcv-full`` 1.2.1 (``ConvModule`` = conv(bias=False) -> norm -> act), so the
"published algorithm" restated here is torch's conv2d / batch_norm / max_pool2d /
linear / normalize / topk / softmax / interpolate, composed exactly as the
reference composes them.  Each function cites the reference file:line it follows.

Pinning: ``tests/golden/*.npz`` hold outputs of the REAL reference modules
(imported from /root/reference in the build container by
``tests/golden/gen_golden.py``); ``tests/test_oracle_golden.py`` checks this
file against them.  The reference's own tests hold no value assertions for
this path, so those captured vectors are the pin.

``emulate_bf16=True`` additionally rounds tensors to bfloat16 at the points the
HIP path stores bf16 (weights, conv outputs, activations, activation
gradients) so kernel-vs-oracle comparisons are not dominated by storage
rounding; the fp32 mode is the one pinned against the goldens.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# bf16 storage emulation
# --------------------------------------------------------------------------


def round_bf16(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bfloat16, returned as fp32."""
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundBoth(torch.autograd.Function):
    """bf16 rounding of a stored activation: value forward, gradient backward."""

    @staticmethod
    def forward(ctx, x):
        return round_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return round_bf16(g)


class _RoundFwd(torch.autograd.Function):
    """bf16 rounding of a weight copy: gradients to the fp32 master stay fp32."""

    @staticmethod
    def forward(ctx, x):
        return round_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


def _act_round(x, on):
    return _RoundBoth.apply(x) if on else x


class _BNGivenStats(torch.autograd.Function):
    """y = gamma * (x - mean) * invstd + beta with GIVEN batch statistics; backward = the batch-norm formula evaluated with x^ from
    x and those statistics.  bf16-storage emulation only: the HIP engine takes the statistics of most conv outputs from the GEMM
    epilogue's fp32 accumulators (vfs_amd/engine.py conv_fwd: `fused` rows) and normalises the bf16-STORED tensor with them."""

    @staticmethod
    def forward(ctx, x, mean, invstd, gamma, beta):
        sh = [1, -1] + [1] * (x.dim() - 2)
        xh = (x - mean.view(sh)) * invstd.view(sh)
        ctx.save_for_backward(xh, invstd, gamma)
        return xh * gamma.view(sh) + beta.view(sh)

    @staticmethod
    def backward(ctx, g):
        xh, invstd, gamma = ctx.saved_tensors
        dims = [0] + list(range(2, g.dim()))
        sh = [1, -1] + [1] * (g.dim() - 2)
        n = g.numel() / g.shape[1]
        sg, sgx = g.sum(dims), (g * xh).sum(dims)
        dx = (gamma * invstd).view(sh) * (g - (sg / n).view(sh) - xh * (sgx / n).view(sh))
        return dx, None, None, sgx, sg


def _w_round(w, on):
    return _RoundFwd.apply(w) if on else w


# --------------------------------------------------------------------------
# ResNet  (reference: mmaction/models/backbones/resnet.py)
# --------------------------------------------------------------------------


class ConvBN(nn.Module):
    """mmcv ``ConvModule`` as the reference uses it: conv(bias=False) -> BN ->
    optional ReLU; sub-module names ``conv`` / ``bn`` feed the state_dict keys
    (resnet.py:51-73, 163-191, 267-277, 425-434)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, relu=True, dilation=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(cout)  # eps 1e-5, momentum 0.1 (torch defaults, as mmcv)
        self.relu = relu
        self.emulate_bf16 = False
        # where the bf16-storage emulation takes the batch statistics from (one of several equally valid DRAWS of the rounding
        # noise, tests/test_cfg1_golden.py): 'stored' = the rounded conv output; 'engine' = the fp32 accumulators where the HIP
        # engine has them (rows per BatchNorm batch % 128 == 0 or > 2048), the stored output elsewhere; 'acc' = accumulators always
        self.emulate_stats = 'stored'
        # mmcv ConvModule ctor: kaiming_init(conv) + constant_init(bn, 1)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        nn.init.constant_(self.bn.weight, 1)
        nn.init.constant_(self.bn.bias, 0)

    @property
    def norm(self):  # mmcv exposes the norm layer as ``.norm`` too
        return self.bn

    def raw(self, x):
        e = self.emulate_bf16
        y = F.conv2d(x, _w_round(self.conv.weight, e), None, self.conv.stride, self.conv.padding, self.conv.dilation)
        return _act_round(y, e)

    def raw_bn(self, x):
        """bn(raw(x)); with emulate_stats != 'stored' (bf16 emulation, training) the batch statistics come from the UNROUNDED conv
        output while the rounded one is normalised (running statistics are not updated in that mode: gradient studies only)"""
        e = self.emulate_bf16
        if not (e and self.bn.training and self.emulate_stats != 'stored'):
            return self.bn(self.raw(x))
        yu = F.conv2d(x, _w_round(self.conv.weight, e), None, self.conv.stride, self.conv.padding, self.conv.dilation)
        yr = _act_round(yu, e)
        rows = yu.shape[0] * yu.shape[2] * yu.shape[3]
        if self.emulate_stats == 'engine' and not (rows % 128 == 0 or rows > 2048):
            return self.bn(yr)
        with torch.no_grad():
            d = yu.detach().double()
            mean, var = d.mean([0, 2, 3]), d.var([0, 2, 3], unbiased=False)
            invstd = (1.0 / torch.sqrt(var + self.bn.eps)).float()
        return _BNGivenStats.apply(yr, mean.float(), invstd, self.bn.weight, self.bn.bias)

    def forward(self, x, residual=None):
        e = self.emulate_bf16
        y = self.raw_bn(x)
        if residual is not None:
            y = y + residual
        if self.relu or residual is not None:
            y = F.relu(y)
        return _act_round(y, e)

    def forward_noact(self, x):
        """conv -> bn without rounding/activation (downsample branch: the HIP
        path keeps the raw conv output and applies this BN inside the consumer)."""
        return self.raw_bn(x)


class BasicBlock(nn.Module):
    """resnet.py:15-113: relu(bn2(conv2(relu(bn1(conv1(x))))) + identity)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, style='pytorch'):      # (style: Bottleneck only)
        super().__init__()
        self.conv1 = ConvBN(inplanes, planes, 3, stride, dilation, relu=True, dilation=dilation)   # resnet.py:51-58
        self.conv2 = ConvBN(planes, planes, 3, 1, 1, relu=False)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample.forward_noact(x)
        out = self.conv1(x)
        return self.conv2(out, residual=identity)


class Bottleneck(nn.Module):
    """resnet.py:116-232, style='pytorch': the stride sits on the 3x3."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, style='pytorch'):
        super().__init__()
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)                                 # resnet.py:156-161
        self.conv1 = ConvBN(inplanes, planes, 1, s1, 0, relu=True)
        self.conv2 = ConvBN(planes, planes, 3, s2, dilation, relu=True, dilation=dilation)         # resnet.py:172-179
        self.conv3 = ConvBN(planes, planes * 4, 1, 1, 0, relu=False)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample.forward_noact(x)
        out = self.conv2(self.conv1(x))
        return self.conv3(out, residual=identity)


class ResNet(nn.Module):
    """resnet.py:309-654 (depth 18/34/50/101/152, style pytorch)."""
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)),
                     50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)),
                     152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, strides=(1, 2, 2, 2), out_indices=(3,), zero_init_residual=True,
                 stop_after_out=False, num_stages=4, dilations=(1, 1, 1, 1), style='pytorch'):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        block, stage_blocks = self.arch_settings[depth]
        self.depth, self.strides, self.out_indices = depth, tuple(strides), tuple(out_indices)
        self.stop_after_out = stop_after_out
        self.conv1 = ConvBN(3, 64, 7, 2, 3, relu=True)           # resnet.py:422-434
        self.maxpool = nn.MaxPool2d(3, 2, 1)                      # resnet.py:435
        inplanes = 64
        self.res_layers = []
        for i, nb in enumerate(stage_blocks[:num_stages]):
            planes, stride = 64 * 2 ** i, strides[i]
            down = None
            if stride != 1 or inplanes != planes * block.expansion:   # resnet.py:266-277
                down = ConvBN(inplanes, planes * block.expansion, 1, stride, 0, relu=False)
            dil = dilations[i]                                            # resnet.py:279-300
            layers = [block(inplanes, planes, stride, down, dil if dil == 1 else dil // 2, style=style)]
            inplanes = planes * block.expansion
            layers += [block(inplanes, planes, 1, None, dil, style=style) for _ in range(1, nb)]
            self.add_module(f'layer{i + 1}', nn.Sequential(*layers))
            self.res_layers.append(f'layer{i + 1}')
        self.feat_dim = inplanes
        if zero_init_residual:                                     # resnet.py:541-551
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.conv3.bn.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.conv2.bn.weight, 0)

    def forward(self, x):                                         # resnet.py:555-575
        x = self.maxpool(self.conv1(x))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
            if self.stop_after_out and i >= max(self.out_indices):
                break  # the reference runs the remaining stages and discards them
        return outs[0] if len(outs) == 1 else tuple(outs)


# --------------------------------------------------------------------------
# SimSiam head + cosine loss
# --------------------------------------------------------------------------


class _Linear(nn.Linear):
    emulate_bf16 = False

    def forward(self, x):
        e = self.emulate_bf16
        return _act_round(F.linear(x, _w_round(self.weight, e), self.bias), e)


class _BN1d(nn.BatchNorm1d):
    emulate_bf16 = False
    relu = False

    def forward(self, x):
        y = super().forward(x)
        if self.relu:
            y = F.relu(y)
        return _act_round(y, self.emulate_bf16)


class SimSiamHead(nn.Module):
    """heads/sim_siam_head.py:28-163 with the configs' settings: avg-pool,
    projector 3x(Linear+BN[+ReLU]) and predictor Linear+BN+ReLU, Linear; the
    Sequential indices ({0,1,3,4,6,7} / {0,1,3}) match the reference's
    state_dict because its ReLUs occupy slots 2,5 / 2."""

    def __init__(self, in_channels, num_projection_fcs=3, projection_mid_channels=2048,
                 projection_out_channels=2048, num_predictor_fcs=2, predictor_mid_channels=512,
                 predictor_out_channels=2048):
        super().__init__()
        last = in_channels
        proj = []
        for i in range(num_projection_fcs):
            is_last = i == num_projection_fcs - 1
            out = projection_out_channels if is_last else projection_mid_channels
            proj += [_Linear(last, out), _BN1d(out)]
            if not is_last:
                proj[-1].relu = True
                proj.append(nn.Identity())  # slot of the reference's nn.ReLU
            last = out
        self.projection_fcs = nn.Sequential(*proj)
        pred = []
        for i in range(num_predictor_fcs):
            is_last = i == num_predictor_fcs - 1
            out = predictor_out_channels if is_last else predictor_mid_channels
            pred.append(_Linear(last, out))
            if not is_last:
                bn = _BN1d(out)
                bn.relu = True
                pred += [bn, nn.Identity()]
            last = out
        self.predictor_fcs = nn.Sequential(*pred)
        self.emulate_bf16 = False

    def forward(self, x):                                       # sim_siam_head.py:143-163
        x = F.adaptive_avg_pool2d(x, 1).flatten(1)
        x = _act_round(x, self.emulate_bf16)
        z = self.projection_fcs(x)
        p = self.predictor_fcs(z)
        return z, p


def cosine_sim_loss(p, z, negative=False):
    """losses/sim_loss.py:42-63 (with_norm=True, pairwise=False) times
    loss_weight 1.0 (losses/base.py:25-37)."""
    p = F.normalize(p, p=2, dim=1)
    z = F.normalize(z, p=2, dim=1)
    prod = torch.sum(p * z, dim=1).view(p.size(0), -1)
    return -prod.mean(dim=-1) if negative else 2 - 2 * prod.mean(dim=-1)


def cosine_sim_loss_general(a, l, mask=None, with_norm=True, negative=False, pairwise=False, loss_weight=1.0):
    """losses/sim_loss.py:42-63 with every constructor option, losses/base.py:25-37: operands [B,C,*]; pairwise: the affinity
    matrix A^T L per sample (a sum over C for every pair of positions), optionally masked, averaged over all pairs."""
    if with_norm:
        a = a / a.norm(dim=1, keepdim=True).clamp_min(1e-12)
        l = l / l.norm(dim=1, keepdim=True).clamp_min(1e-12)
    if mask is not None:
        assert pairwise
    B = a.shape[0]
    if pairwise:
        a3, l3 = a.reshape(B, a.shape[1], -1), l.reshape(B, l.shape[1], -1)      # flatten(2) fails on [N,C] in the reference too
        if a.ndim < 3:
            raise IndexError('Dimension out of range')
        prod = torch.bmm(a3.transpose(1, 2), l3)
        if mask is not None:
            assert prod.shape == mask.shape
            prod = prod * mask.float()
        prod = prod.reshape(B, -1)
    else:
        prod = (a * l).sum(dim=1).reshape(B, -1)
    m = prod.mean(dim=-1)
    return (-m if negative else 2 - 2 * m) * loss_weight


def head_loss(p1, z1, p2, z2, weight=1.0):
    """heads/sim_siam_head.py:165-174: symmetric, stop-gradient on z."""
    return (cosine_sim_loss(p1, z2.detach()) * 0.5 + cosine_sim_loss(p2, z1.detach()) * 0.5) * weight


def video2images(imgs):                                          # common/utils.py:45-53
    b, c, t = imgs.shape[:3]
    if t == 1:
        return imgs.squeeze(2).reshape(b, c, *imgs.shape[3:])
    return imgs.transpose(1, 2).contiguous().reshape(b * t, c, *imgs.shape[3:])


def images2video(imgs, clip_len):                                # common/utils.py:56-64
    b, c = imgs.shape[:2]
    if clip_len == 1:
        return imgs.unsqueeze(2)
    return imgs.reshape(b // clip_len, clip_len, c, *imgs.shape[2:]).transpose(1, 2).contiguous()


class SimSiamTracker(nn.Module):
    """trackers/sim_siam_base_tracker.py:12-76 + trackers/base.py:76-156."""

    def __init__(self, depth, head_kwargs, intra_video=False, **backbone_kwargs):
        super().__init__()
        self.backbone = ResNet(depth, **backbone_kwargs)
        self.img_head = SimSiamHead(**head_kwargs)
        self.intra_video = intra_video
        self.register_buffer('iteration', torch.tensor(0, dtype=torch.float))  # base.py:39

    def set_emulate_bf16(self, on=True, stats='stored'):
        assert stats in ('stored', 'engine', 'acc')
        for m in self.modules():
            if hasattr(m, 'emulate_bf16'):
                m.emulate_bf16 = on
            if hasattr(m, 'emulate_stats'):
                m.emulate_stats = stats
        return self

    def forward_img_head(self, x1, x2, clip_len):                # sim_siam_base_tracker.py:31-56
        losses = OrderedDict()
        z1, p1 = self.img_head(x1)
        z2, p2 = self.img_head(x2)
        w = 1.0 / clip_len if self.intra_video else 1.0
        losses['0.loss_feat'] = head_loss(p1, z1, p2, z2, w)
        if self.intra_video:
            z2v, p2v = images2video(z2, clip_len), images2video(p2, clip_len)
            for i in range(1, clip_len):
                losses[f'{i}.loss_feat'] = head_loss(
                    p1, z1, video2images(p2v.roll(i, dims=2)), video2images(z2v.roll(i, dims=2)), w)
        return losses

    def forward_train(self, imgs):                               # sim_siam_base_tracker.py:58-76
        assert imgs.size(1) == 2 and imgs.ndim == 6
        clip_len = imgs.size(3)
        imgs1 = video2images(imgs[:, 0].contiguous().reshape(-1, *imgs.shape[2:]))
        imgs2 = video2images(imgs[:, 1].contiguous().reshape(-1, *imgs.shape[2:]))
        if getattr(self.backbone.conv1, 'emulate_bf16', False):
            imgs1, imgs2 = round_bf16(imgs1), round_bf16(imgs2)
        x1 = self.backbone(imgs1)
        x2 = self.backbone(imgs2)
        losses = self.forward_img_head(x1, x2, clip_len)
        return OrderedDict((f'img_head.{k}', v) for k, v in losses.items())


def parse_losses(losses):
    """trackers/base.py:76-110 without the distributed all-reduce."""
    log_vars = OrderedDict((k, v.mean()) for k, v in losses.items())
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, OrderedDict((k, float(v.detach())) for k, v in log_vars.items())


def sgd_step(params, grads, bufs, lr, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD as configs/*:134 sets it (dampening 0, nesterov False):
    g += wd*w ; buf = g (first step) or mom*buf + g ; w -= lr*buf."""
    for i, (p, g) in enumerate(zip(params, grads)):
        g = g + weight_decay * p
        bufs[i] = g.clone() if bufs[i] is None else momentum * bufs[i] + g
        p.sub_(lr * bufs[i])


# --------------------------------------------------------------------------
# DAVIS label propagation  (vanilla_tracker.py, common/local_attention.py)
# --------------------------------------------------------------------------


def pil_nearest_resize(label: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """common/utils.py:25-42 resizes the uint8 label map with PIL NEAREST via
    mmcv.imresize; PIL samples src = floor((dst + 0.5) * in/out)."""
    in_h, in_w = label.shape
    ys = np.minimum(np.floor((np.arange(out_h) + 0.5) * (in_h / out_h)).astype(np.int64), in_h - 1)
    xs = np.minimum(np.floor((np.arange(out_w) + 0.5) * (in_w / out_w)).astype(np.int64), in_w - 1)
    return label[ys][:, xs]


def spatial_neighbor_circle(height, width, neighbor_range, dtype=torch.float32):
    """common/affinity_utils.py:144-156: M[i,j] = ||pos_i - pos_j||_2 < range//2."""
    radius = neighbor_range // 2
    gy, gx = torch.meshgrid(torch.arange(height, dtype=dtype), torch.arange(width, dtype=dtype),
                            indexing='ij')
    d = ((gy.view(height, width, 1, 1) - gy.view(1, 1, height, width)) ** 2 +
         (gx.view(height, width, 1, 1) - gx.view(1, 1, height, width)) ** 2) ** 0.5
    return (d < radius).view(height * width, height * width)


def masked_attention_efficient(query, key, value, mask, temperature=1.0, topk=None, normalize=True,
                               step=32, dtype=None):
    """common/local_attention.py:237-348 (mode 'softmax', non_mask_len 0).
    query [N,C,H,W], key [N,C,T,H,W], value [N,Co,T,H,W], mask [HW,HW] bool."""
    if dtype is not None:
        query, key, value = query.to(dtype), key.to(dtype), value.to(dtype)
    n, c, qh, qw = query.shape
    t, kh, kw = key.shape[2:]
    co = value.size(1)
    if normalize:
        query = F.normalize(query, p=2, dim=1)
        key = F.normalize(key, p=2, dim=1)
    q = query.reshape(n, c, qh * qw)
    k = key.reshape(n, c, t * kh * kw)
    v = value.reshape(n, co, t * kh * kw)
    out = torch.zeros(n, co, qh * qw, dtype=query.dtype)
    for ptr in range(0, qh * qw, step):
        aff = torch.einsum('bci,bcj->bij', k, q[..., ptr:ptr + step]) / temperature
        if mask is not None:
            cur = mask.view(1, 1, kh * kw, qh * qw)[..., ptr:ptr + step].expand(n, t, -1, -1)
            aff.masked_fill_(~cur.reshape(n, -1, aff.size(2)), float('-inf'))
        if topk is not None:
            tv, ti = aff.topk(k=topk, dim=1)
            sel = v.transpose(0, 1).reshape(co, -1).index_select(1, ti.reshape(-1))
            sel = sel.reshape(co, *ti.shape).transpose(0, 1)
            cur_out = torch.einsum('bcks,bks->bcs', sel, tv.softmax(dim=1))
        else:
            cur_out = torch.einsum('bck,bks->bcs', v, aff.softmax(dim=1))
        out[..., ptr:ptr + step] = cur_out
    return out.reshape(n, co, qh, qw)


def seg_postprocess(seg_logit, out_hw):
    """vanilla_tracker.py:162-181: bilinear upsample (align_corners=False),
    per-channel min-max normalisation where max > 0, argmax -> uint8."""
    sp = F.interpolate(seg_logit, size=out_hw, mode='bilinear', align_corners=False)
    mn = sp.flatten(2).min(dim=-1)[0].view(*sp.shape[:2], 1, 1)
    mx = sp.flatten(2).max(dim=-1)[0].view(*sp.shape[:2], 1, 1)
    norm = (sp - mn) / (mx - mn + 1e-12)
    sp = torch.where(mx > 0, norm, sp)
    return sp.argmax(dim=1).to(torch.uint8)


def label_propagate(feats, ref_seg_map, out_hw, *, precede_frames=20, topk=10, temperature=0.07,
                    neighbor_range=None, with_first=True, dtype=None, return_logits=False, normalize=True):
    """vanilla_tracker.py:80-206 given the feature bank ``feats`` [1,C,T,h,w]
    (what get_feats returns) and the first-frame uint8 labels [H,W].
    Returns uint8 [T,H_out,W_out] (frame 0 = nearest-resized ground truth)."""
    _, c, clip_len, h, w = feats.shape
    small = pil_nearest_resize(ref_seg_map, h, w)
    seg0 = F.one_hot(torch.from_numpy(small.astype(np.int64))[None]).permute(0, 3, 1, 2).float()
    ref_out = F.interpolate(torch.from_numpy(ref_seg_map)[None, None].float(), size=out_hw,
                            mode='nearest')[0, 0].to(torch.uint8)
    mask = spatial_neighbor_circle(h, w, neighbor_range) if neighbor_range is not None else None
    seg_bank, preds, logits = [seg0], [ref_out], []
    for f in range(1, clip_len):
        ks = max(0, f - precede_frames)
        q = feats[:, :, f]
        k = feats[:, :, ks:f]
        v = torch.stack(seg_bank[ks:f], dim=2)
        if with_first:   # frame 0 is prepended even when it is already in the window
            k = torch.cat([feats[:, :, 0:1], k], dim=2)
            v = torch.cat([seg_bank[0].unsqueeze(2), v], dim=2)
        seg = masked_attention_efficient(q, k, v, mask, temperature, topk, normalize, dtype=dtype).float()
        seg_bank.append(seg)
        logits.append(seg)
        preds.append(seg_postprocess(seg, out_hw)[0])
    out = torch.stack(preds, 0).numpy()
    return (out, logits) if return_logits else out


class VanillaTracker(nn.Module):
    """trackers/vanilla_tracker.py:16-206 for the configs' test_cfg."""

    def __init__(self, depth, test_cfg):
        super().__init__()
        self.test_cfg = dict(test_cfg)
        self.backbone = ResNet(depth, strides=test_cfg.get('strides', (1, 2, 1, 1)),
                               out_indices=test_cfg.get('out_indices', (2,)), stop_after_out=True)

    def extract_feat_test(self, frames):
        """vanilla_tracker.py:30-46: with test_cfg.all_blocks the output of EVERY residual block of the
        stages in test_cfg.out_indices (a tuple), otherwise the backbone's stage output."""
        if not self.test_cfg.get('all_blocks', False):
            return self.backbone(frames)
        bb = self.backbone
        x = bb.maxpool(bb.conv1(frames))
        outs = []
        stages = tuple(self.test_cfg.get('out_indices', (2,)))
        for i, name in enumerate(bb.res_layers):
            layer = getattr(bb, name)
            if i in stages:
                for block in layer:
                    x = block(x)
                    outs.append(x)
            else:
                x = layer(x)
        return tuple(outs)

    @torch.no_grad()
    def forward_test(self, imgs, ref_seg_map, original_shape):
        imgs = imgs.reshape((-1,) + imgs.shape[2:])           # [1,3,T,H,W]
        frames = video2images(imgs)
        step = self.test_cfg.get('batch_step', 10)
        chunks = [self.extract_feat_test(frames[i:i + step]) for i in range(0, frames.size(0), step)]
        tc = self.test_cfg
        many = isinstance(chunks[0], tuple)
        res = []
        for fi in range(len(chunks[0]) if many else 1):
            feats = torch.cat([c[fi] if many else c for c in chunks])
            feats = images2video(feats, frames.size(0))
            res.append(label_propagate(feats, ref_seg_map, tuple(original_shape[:2]),
                                       precede_frames=tc['precede_frames'], topk=tc['topk'],
                                       temperature=tc['temperature'], neighbor_range=tc.get('neighbor_range'),
                                       with_first=tc.get('with_first', True)))
        # vanilla_tracker.py:199-205: several feature levels are stacked along axis 1 of [1, ...]
        return np.stack(res, axis=0) if many else res[0]


# --------------------------------------------------------------------------
# deterministic closed-form fillers (shared by gen_golden.py and the tests so
# fixtures only need to store outputs)
# --------------------------------------------------------------------------


def fill_tensor(shape, seed, scale=1.0, offset=0.0):
    """Closed-form pseudo-random fill in [-scale, scale) + offset, independent of
    torch's RNG implementation: frac(sin(i*12.9898 + seed*78.233) * 43758.5453)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * 12.9898 + (seed + 1) * 78.233) * 43758.5453
    v = (v - np.floor(v)) * 2.0 - 1.0
    return torch.from_numpy((v * scale + offset).astype(np.float32).reshape(shape))


def fill_state_dict_(module: nn.Module, seed: int = 0):
    """Non-degenerate deterministic weights for every parameter/buffer, keyed by
    a CRC of the state_dict NAME (BN gamma around 1 incl. the zero-init ones, running
    stats non-trivial) so conv bugs cannot hide behind zero-init residuals."""
    sd = module.state_dict()
    with torch.no_grad():
        for idx, (name, t) in enumerate(sd.items()):
            if name.endswith('num_batches_tracked') or name == 'iteration':
                continue
            s = seed * 100003 + zlib.crc32(name.encode()) % 100000
            if name.endswith('running_var'):
                t.copy_(fill_tensor(t.shape, s, 0.3, 1.0))
            elif name.endswith('running_mean'):
                t.copy_(fill_tensor(t.shape, s, 0.2))
            elif '.bn.' in name or (t.ndim == 1 and name.endswith('weight')):
                if name.endswith('weight'):
                    t.copy_(fill_tensor(t.shape, s, 0.3, 1.0))
                else:
                    t.copy_(fill_tensor(t.shape, s, 0.2))
            elif name.endswith('bias'):
                t.copy_(fill_tensor(t.shape, s, 0.1))
            else:
                fan_in = int(np.prod(t.shape[1:])) if t.ndim > 1 else t.numel()
                t.copy_(fill_tensor(t.shape, s, math.sqrt(3.0 / fan_in) * 1.4))
    return module


HEAD_KW = {18: dict(in_channels=512, projection_mid_channels=512, projection_out_channels=512,
                    predictor_mid_channels=128, predictor_out_channels=512),
           50: dict(in_channels=2048, projection_mid_channels=2048, projection_out_channels=2048,
                    predictor_mid_channels=512, predictor_out_channels=2048)}


def build_tracker(depth, intra_video=None, head_kw=None, **backbone_kwargs):
    """The two shipped model configs (configs/r18_*:2-26, configs/r50_*:2-26); head_kw /
    backbone_kwargs allow truncated variants for cheap tests."""
    if intra_video is None:
        intra_video = depth == 18
    return SimSiamTracker(depth, head_kw or HEAD_KW[depth], intra_video, **backbone_kwargs)
