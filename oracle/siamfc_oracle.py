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
