"""The SiamFC linear probe on a frozen VFS backbone (OTB-100 metric of the reference, README.md:87-90): the training and
tracking logic of `TrackerSiamFC` (projects/siamfc-pytorch/siamfc/siamfc_tracker_base.py:88-500) on the HIP kernels, without
its third-party scaffolding (got10k's Tracker base / dataset classes, cv2, torchvision).

  train_step   (:364-387)  frozen dilated backbone (eval) -> 1x1 convs -> cross-correlation -> Balanced / Focal loss ->
                           backward through the head (vfs_xcorr_bwd, vfs_conv_wgrad, vfs_bias_grad) -> Adam / SGD
  _create_labels (:456-500), current_lr (:349-362), the optimizer / scheduler choices of __init__ (:131-166)
  init / update / track (:199-347): the tracking loop.  Crops and the response up-sampling are cv2 calls in the reference
                           (ops.crop_and_resize -> image_utils.get_cropped_input, cv2.resize INTER_CUBIC); cv2 is absent in
                           this image, so `crop_and_resize` / `resize_cubic` below restate the published OpenCV arithmetic -
                           parity UNPINNED for those two functions (said in DESIGN.md); everything downstream of them is
                           pinned by tests/golden/siamfc_*.npz.
Only num_convs=1, kernel_size=1 heads train here (what the probe's config builds, default_config_base.py:33-37)."""
import numpy as np
import torch
import torch.nn as nn

from .engine import BF16, bump_params_epoch, shared_engine
from .packing import wgrad_splits
from .siamfc_heads import SiamConvFC, SiamFC, _nhwc_bf16

DEFAULT_CFG = dict(      # default_config_base.py:2-51
    out_scale=0.001, exemplar_sz=120, instance_sz=255, context=0.5, scale_num=3, scale_step=1.0375, scale_lr=0.59, scale_penalty=0.9745,
    window_influence=0.176, response_sz=17, response_up=16, total_stride=8, epoch_num=50, batch_size=8, initial_lr=1e-3,
    ultimate_lr=1e-5, weight_decay=5e-4, momentum=0.9, r_pos=16, r_neg=0, optimizer='Adam', loss='focal', lr_schedule='exp',
    lr_step_size=10, extra_conv=True, out_channels=512, reduction=1, force_wd=False,
    backbone=dict(frozen_stages=4, dilations=(1, 1, 2, 4), strides=(1, 2, 1, 1), out_indices=(3,), norm_eval=True))
MEAN = (123.675, 116.28, 103.53)
STD = (58.395, 57.12, 57.375)


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam (amsgrad off) through vfs_adam_step, one launch per parameter tensor (the probe has four)"""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        eng = shared_engine()
        bump_params_epoch()      # raw-pointer update: packed / folded copies of the parameters must refresh
        for grp in self.param_groups:
            for p in grp['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'], st['exp_avg_sq'] = torch.zeros_like(p), torch.zeros_like(p)
                st['step'] += 1
                eng.lib.adam_step(p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], p.numel(), float(grp['lr']), float(grp['betas'][0]),
                                  float(grp['betas'][1]), float(grp['eps']), float(grp['weight_decay']), int(st['step']),
                                  eng.stream(p.device))


class ParamSGD(torch.optim.Optimizer):
    """torch.optim.SGD (momentum, weight decay) through vfs_sgd_step, one launch per parameter tensor"""

    def __init__(self, params, lr=1e-2, momentum=0.9, weight_decay=0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        eng = shared_engine()
        bump_params_epoch()      # raw-pointer update: packed / folded copies of the parameters must refresh
        for grp in self.param_groups:
            for p in grp['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if 'momentum_buffer' not in st:
                    st['momentum_buffer'] = torch.zeros_like(p)
                eng.lib.sgd_step(p.data, p.grad, st['momentum_buffer'], p.numel(), float(grp['lr']), float(grp['momentum']),
                                 float(grp['weight_decay']), None, eng.stream(p.device))


def create_labels(size, r_pos, r_neg, total_stride, device):
    """siamfc_tracker_base.py:456-500: logistic labels on the response grid (block distance to the centre)"""
    n, c, h, w = size
    x, y = np.meshgrid(np.arange(w) - (w - 1) / 2, np.arange(h) - (h - 1) / 2)
    dist = np.abs(x) + np.abs(y)
    lab = np.where(dist <= r_pos / total_stride, 1.0, np.where(dist < r_neg / total_stride, 0.5, 0.0))
    return torch.from_numpy(np.tile(lab.reshape(1, 1, h, w), (n, c, 1, 1))).float().to(device)


def head_loss_backward(head, zf, xf, labels, loss='focal', param=None, backward=True):
    """responses = head(z, x); loss (losses.py: 'balance' | 'focal'); with backward=True the head's parameter gradients are
    ACCUMULATED into .grad (create them with zero_grad first).  zf / xf: NCHW fp32 features.  -> (loss tensor [1], responses)"""
    eng = shared_engine(xf.device)
    dev = xf.device
    s = eng.stream(dev)
    z, x = _nhwc_bf16(zf), _nhwc_bf16(xf)
    convs = isinstance(head, SiamConvFC)
    if convs and (len(head.z_convs) != 1):
        raise NotImplementedError('probe training: num_convs=1 heads (siamfc_tracker_base.py:108-113)')
    zc, xc = (head._conv1x1(head.z_convs[0], z), head._conv1x1(head.x_convs[0], x)) if convs else (z, x)
    nz, hz, wz, c = zc.shape
    nx, h, w, _ = xc.shape
    ho, wo = h - hz + 1, w - wz + 1
    resp = torch.empty(nx, 1, ho, wo, device=dev)
    eng.lib.xcorr_fwd(zc, xc, resp, nz, nx, hz, wz, h, w, c, float(head.out_scale), s)
    mode = {'balance': 0, 'focal': 1}[loss]
    if param is None:
        param = 1.0 if mode == 0 else 2.0
    out = torch.empty(1, device=dev)
    g = torch.empty_like(resp) if backward else None
    eng.lib.siamfc_loss(resp, labels.contiguous(), out, g, resp.numel(), mode, float(param), 1.0, s)
    if backward and convs:
        dzc, dxc = torch.empty_like(zc), torch.empty_like(xc)
        eng.lib.xcorr_bwd(zc, xc, g, dzc, dxc, nz, nx, hz, wz, h, w, c, float(head.out_scale), s)
        for conv, feat, d in ((head.z_convs[0], z, dzc), (head.x_convs[0], x, dxc)):
            n_, hh, ww, cin = feat.shape
            M, cout = n_ * hh * ww, conv.out_channels
            nsplit, pps = wgrad_splits(M, cout, cin)
            partial = eng.ws('ws.wgrad', nsplit * cout * cin, torch.float32, dev)
            if conv.weight.grad is None:
                conv.weight.grad = torch.zeros_like(conv.weight)
            eng.lib.conv_wgrad(d, feat, partial, conv.weight.grad, n_, hh, ww, cin, hh, ww, cout, 1, 1, 1, 0, nsplit, pps, s)
            if conv.bias is not None:
                if conv.bias.grad is None:
                    conv.bias.grad = torch.zeros_like(conv.bias)
                eng.lib.bias_grad(d, conv.bias.grad, M, cout, s)
    return out, resp


class SiamFCProbe:
    """TrackerSiamFC (siamfc_tracker_base.py:88-500): `cfg` = the reference's default_cfg keys (DEFAULT_CFG above)."""

    def __init__(self, cfg=None, depth=50, backbone=None, device=None):
        from .resnet import ResNet
        self.cfg = dict(DEFAULT_CFG, **(cfg or {}))
        c = self.cfg
        self.device = torch.device(device) if device is not None else torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')
        self.backbone = backbone if backbone is not None else ResNet(depth, norm_cfg=dict(type='BN', requires_grad=True), **c['backbone'])
        self.head = (SiamConvFC(c['out_channels'], c['out_channels'] // c['reduction'], out_scale=c['out_scale']) if c['extra_conv']
                     else SiamFC(out_scale=c['out_scale']))
        self.backbone.to(self.device).eval()          # frozen_stages=4 + norm_eval: train(True) leaves every layer in eval mode
        self.head.to(self.device)
        params = [p for p in self.head.parameters() if p.requires_grad]
        wd = c['weight_decay'] if (c['backbone'].get('frozen_stages', -1) < 4 or c['force_wd']) else 0     # :135-137
        if not params:
            self.optimizer = None
        elif c['optimizer'] == 'SGD':
            self.optimizer = ParamSGD(params, lr=c['initial_lr'], weight_decay=wd, momentum=c['momentum'])
        elif c['optimizer'] == 'Adam':
            self.optimizer = Adam(params, lr=c['initial_lr'], weight_decay=wd)
        else:
            raise NotImplementedError(c['optimizer'])
        self.lr_scheduler = None
        if self.optimizer is not None:
            if c['lr_schedule'] == 'exp':
                gamma = np.power(c['ultimate_lr'] / c['initial_lr'], 1.0 / c['epoch_num'])
                self.lr_scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma)
            elif c['lr_schedule'] == 'step':
                self.lr_scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, c['lr_step_size'])
            elif c['lr_schedule'] != 'fixed':
                raise NotImplementedError(c['lr_schedule'])
        if c['loss'] not in ('balance', 'focal'):
            raise NotImplementedError(c['loss'])
        self._mean = torch.tensor(MEAN, device=self.device).view(1, 3, 1, 1)
        self._std = torch.tensor(STD, device=self.device).view(1, 3, 1, 1)
        self.labels = None

    # ------------------------------------------------------------------ training
    def normalize(self, imgs):
        return (imgs - self._mean) / self._std

    def features(self, imgs):
        """frozen backbone, eval mode (reference precision: the fp32 evaluation path)"""
        with torch.no_grad():
            return self.backbone(self.normalize(imgs.to(self.device).float()))

    def _create_labels(self, size):
        if self.labels is not None and tuple(self.labels.size()) == tuple(size):
            return self.labels
        c = self.cfg
        self.labels = create_labels(size, c['r_pos'], c['r_neg'], c['total_stride'], self.device)
        return self.labels

    def current_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def train_step(self, batch, backward=True):
        """batch = (z [N,3,ez,ez], x [N,3,ix,ix]) uint8-range RGB floats, as the reference's Pair dataset yields them"""
        zf, xf = self.features(batch[0]), self.features(batch[1])
        nz, _, hz, wz = zf.shape
        nx, _, h, w = xf.shape
        labels = self._create_labels((nx, 1, h - hz + 1, w - wz + 1))
        if backward and self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=False)
            for p in self.head.parameters():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
        loss, _ = head_loss_backward(self.head, zf, xf, labels, self.cfg['loss'], backward=backward and self.optimizer is not None)
        if backward and self.optimizer is not None:
            self.optimizer.step()
        return float(loss.item())

    # ------------------------------------------------------------------ tracking (:199-347)
    @torch.no_grad()
    def init(self, img, box):
        c = self.cfg
        box = np.array([box[1] - 1 + (box[3] - 1) / 2, box[0] - 1 + (box[2] - 1) / 2, box[3], box[2]], dtype=np.float32)
        self.center, self.target_sz = box[:2], box[2:]
        self.upscale_sz = c['response_up'] * c['response_sz']
        self.hann_window = np.outer(np.hanning(self.upscale_sz), np.hanning(self.upscale_sz))
        self.hann_window /= self.hann_window.sum()
        self.scale_factors = c['scale_step'] ** np.linspace(-(c['scale_num'] // 2), c['scale_num'] // 2, c['scale_num'])
        context = c['context'] * np.sum(self.target_sz)
        self.z_sz = np.sqrt(np.prod(self.target_sz + context))
        self.x_sz = self.z_sz * c['instance_sz'] / c['exemplar_sz']
        self.avg_color = np.mean(img, axis=(0, 1))
        z = crop_and_resize(img, self.center, self.z_sz, c['exemplar_sz'], self.avg_color)
        z = torch.from_numpy(z).to(self.device).permute(2, 0, 1).unsqueeze(0).float()
        self.kernel = self.features(z)

    @torch.no_grad()
    def update(self, img):
        c = self.cfg
        x = np.stack([crop_and_resize(img, self.center, self.x_sz * f, c['instance_sz'], self.avg_color) for f in self.scale_factors])
        x = torch.from_numpy(x).to(self.device).permute(0, 3, 1, 2).float()
        responses = self.head(self.kernel, self.features(x)).squeeze(1).cpu().numpy()
        responses = np.stack([resize_cubic(u, self.upscale_sz, self.upscale_sz) for u in responses])
        responses[:c['scale_num'] // 2] *= c['scale_penalty']
        responses[c['scale_num'] // 2 + 1:] *= c['scale_penalty']
        scale_id = np.argmax(np.amax(responses, axis=(1, 2)))
        response = responses[scale_id]
        response -= response.min()
        response /= response.sum() + 1e-16
        response = (1 - c['window_influence']) * response + c['window_influence'] * self.hann_window
        loc = np.unravel_index(response.argmax(), response.shape)
        disp_in_response = np.array(loc) - (self.upscale_sz - 1) / 2
        disp_in_instance = disp_in_response * c['total_stride'] / c['response_up']
        disp_in_image = disp_in_instance * self.x_sz * self.scale_factors[scale_id] / c['instance_sz']
        self.center += disp_in_image
        scale = (1 - c['scale_lr']) * 1.0 + c['scale_lr'] * self.scale_factors[scale_id]
        self.target_sz *= scale
        self.z_sz *= scale
        self.x_sz *= scale
        return np.array([self.center[1] + 1 - (self.target_sz[1] - 1) / 2, self.center[0] + 1 - (self.target_sz[0] - 1) / 2,
                         self.target_sz[1], self.target_sz[0]])

    def track(self, frames, box):
        """frames: iterable of RGB uint8 arrays [H,W,3] (the reference reads files with cv2); box: 1-indexed ltwh"""
        boxes = []
        for f, img in enumerate(frames):
            if f == 0:
                self.init(img, box)
                boxes.append(np.asarray(box, dtype=np.float64))
            else:
                boxes.append(self.update(img))
        return np.stack(boxes)


# ---------------------------------------------------------------------------------------------
# cv2 stand-ins (cv2 is not in this image).  Restated from the published OpenCV implementation; parity UNPINNED.
# ---------------------------------------------------------------------------------------------
def _resize_linear_u8(img, ow, oh):
    """cv2.resize(INTER_LINEAR) on 8-bit data: 11-bit fixed-point coefficients, horizontal then vertical pass"""
    ih, iw = img.shape[:2]

    def coeffs(n_in, n_out):
        scale = n_in / n_out
        f = (np.arange(n_out) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        fr = f - i0
        lo = i0 < 0
        fr[lo], i0[lo] = 0.0, 0
        hi = i0 >= n_in - 1
        fr[hi] = 0.0
        i0 = np.minimum(i0, n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        c1 = np.rint(fr * 2048).astype(np.int64)
        return i0, i1, 2048 - c1, c1
    x0, x1, a0, a1 = coeffs(iw, ow)
    y0, y1, b0, b1 = coeffs(ih, oh)
    src = img.astype(np.int64)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]            # [ih, ow, c], scaled by 2^11
    out = ((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16)
    return np.clip((out + 2) >> 2, 0, 255).astype(np.uint8)


def crop_and_resize(img, center, size, out_size, border_value):
    """ops.crop_and_resize(faster=True) -> image_utils.get_cropped_input: the in-image part of the square box is resized to
    its share of out_size, the rest is padded with the average colour"""
    size = max(2, float(size))
    yc, xc = float(center[0]), float(center[1])
    box = np.round(np.array([xc - size / 2, yc - size / 2, xc + size / 2, yc + size / 2])).astype(int)     # x0, y0, x1, y1
    wh = np.array([box[2] - box[0], box[3] - box[1]])
    H, W = img.shape[:2]
    patch = img[max(box[1], 0):min(box[3], H), max(box[0], 0):min(box[2], W)]
    out_size = int(out_size)
    fill = np.asarray(border_value, dtype=np.float64)
    if patch.shape[0] == 0 or patch.shape[1] == 0:
        return np.zeros((out_size, out_size, 3), img.dtype)
    bounded = np.clip(box, 0, [W, H, W, H])
    bwh = np.array([bounded[2] - bounded[0], bounded[3] - bounded[1]])
    ow = max(1, int(np.round(out_size * bwh[0] / wh[0])))
    oh = max(1, int(np.round(out_size * bwh[1] / wh[1])))
    patch = _resize_linear_u8(np.ascontiguousarray(patch), ow, oh)
    pad = np.zeros(4, dtype=int)
    pad[:2] = np.maximum(0, -box[:2] * out_size / wh)
    pad[2:] = out_size - (pad[:2] + np.array([patch.shape[1], patch.shape[0]]))
    if np.any(pad != 0):
        if np.any(pad < 0):
            return np.zeros((out_size, out_size, 3))
        full = np.empty((out_size, out_size, 3), patch.dtype)
        full[:] = np.clip(np.rint(fill), 0, 255).astype(patch.dtype) if patch.dtype == np.uint8 else fill
        full[pad[1]:pad[1] + patch.shape[0], pad[0]:pad[0] + patch.shape[1]] = patch
        patch = full
    return patch


def resize_cubic(a, ow, oh):
    """cv2.resize(INTER_CUBIC) of a float32 map: Keys kernel with a = -0.75, pixel centres aligned, replicated border"""
    a = np.asarray(a, dtype=np.float32)

    def taps(n_in, n_out):
        f = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(f).astype(np.int64)
        t = (f - i0).astype(np.float32)
        A = np.float32(-0.75)
        w = np.stack([((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A, ((A + 2) * t - (A + 3)) * t * t + 1,
                      ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1, np.zeros_like(t)], 1).astype(np.float32)
        w[:, 3] = 1 - w[:, 0] - w[:, 1] - w[:, 2]
        idx = np.clip(i0[:, None] + np.arange(-1, 3)[None], 0, n_in - 1)
        return idx, w
    xi, xw = taps(a.shape[1], ow)
    yi, yw = taps(a.shape[0], oh)
    rows = (a[:, xi] * xw[None]).sum(-1)              # [ih, ow]
    return (rows[yi] * yw[:, :, None]).sum(1).astype(np.float32)
