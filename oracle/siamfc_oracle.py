"""CPU restatement (PyTorch fp32) of the SiamFC probe's heads.  TEST INFRASTRUCTURE ONLY.

Follows projects/siamfc-pytorch/siamfc/heads.py: `SiamFC` (:7-23) and `SiamConvFC` (:26-58); pinned against
outputs of the reference classes themselves (tests/golden/siamfc_heads.npz, gen_siamfc_golden.py - the module is pure
torch and imports as it is)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def fast_xcorr(z, x):
    """heads.py:16-23 / 51-58: search feature m is correlated with exemplar m % nz (grouped conv2d)"""
    nz = z.size(0)
    nx, c, h, w = x.size()
    out = F.conv2d(x.view(-1, nz * c, h, w), z, groups=nz)
    return out.view(nx, -1, out.size(-2), out.size(-1))


class SiamFC(nn.Module):
    def __init__(self, out_scale=0.001):
        super().__init__()
        self.out_scale = out_scale

    def forward(self, z, x):
        return fast_xcorr(z, x) * self.out_scale


class SiamConvFC(nn.Module):
    def __init__(self, in_channels, channels, num_convs=1, kernel_size=1, out_scale=0.001):
        super().__init__()
        self.out_scale = out_scale
        zc, xc, last = [], [], in_channels
        for _ in range(num_convs):
            zc.append(nn.Conv2d(last, channels, kernel_size))
            xc.append(nn.Conv2d(last, channels, kernel_size))
            last = channels
        self.z_convs, self.x_convs = nn.Sequential(*zc), nn.Sequential(*xc)

    def forward(self, z, x):
        return fast_xcorr(self.z_convs(z), self.x_convs(x)) * self.out_scale


# ---------------------------------------------------------------------------------------------
# training the probe (siamfc_tracker_base.py:364-387, 456-500; losses.py) - pinned by tests/golden/siamfc_train.npz
# ---------------------------------------------------------------------------------------------
def create_labels(size, r_pos, r_neg, total_stride):
    """siamfc_tracker_base.py:456-500: 1 within block distance r_pos / stride of the centre, 0.5 inside r_neg / stride, else 0"""
    import numpy as np
    n, c, h, w = size
    x = np.arange(w) - (w - 1) / 2
    y = np.arange(h) - (h - 1) / 2
    x, y = np.meshgrid(x, y)
    dist = np.abs(x) + np.abs(y)
    rp, rn = r_pos / total_stride, r_neg / total_stride
    lab = np.where(dist <= rp, np.ones_like(x), np.where(dist < rn, np.ones_like(x) * 0.5, np.zeros_like(x)))
    return torch.from_numpy(np.tile(lab.reshape(1, 1, h, w), (n, c, 1, 1))).float()


def balanced_loss(x, target, neg_weight=1.0):
    """losses.py:24-41"""
    pos, neg = target == 1, target == 0
    w = torch.zeros_like(target)
    w[pos] = 1 / pos.sum().float()
    w[neg] = 1 / neg.sum().float() * neg_weight
    w = w / w.sum()
    return F.binary_cross_entropy_with_logits(x, target, w, reduction='sum')


def focal_loss(x, target, gamma=2):
    """losses.py:44-65 (the normaliser avg_weight.mean() stays in the graph)"""
    ls = torch.clamp(x, max=0) - torch.log(1 + torch.exp(-torch.abs(x)))
    lms = torch.clamp(-x, max=0) - torch.log(1 + torch.exp(-torch.abs(x)))
    p = torch.sigmoid(x)
    pw, nw = torch.pow(1 - p, gamma), torch.pow(p, gamma)
    loss = -(target * pw * ls + (1 - target) * nw * lms)
    avg = target * pw + (1 - target) * nw
    return (loss / avg.mean()).mean()
