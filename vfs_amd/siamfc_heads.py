"""Heads of the SiamFC linear probe (§8f rank 4; projects/siamfc-pytorch/siamfc/heads.py): `SiamFC` (:7-23, plain
cross-correlation) and `SiamConvFC` (:26-58, a 1x1 conv on exemplar and search features first), same constructor
arguments and state_dict names, forward only - the inference side of a trained probe.  The 1x1 convs run through
vfs_conv_fwd, the correlation through vfs_xcorr_fwd (csrc/xcorr.hip).  Training the probe (its losses, Adam) and the
tracker loop (got10k, cv2 crops) are not here."""
import torch
import torch.nn as nn

from .engine import BF16, shared_engine
from .packing import build_pack_table


def _nhwc_bf16(t):
    return t.permute(0, 2, 3, 1).contiguous().to(BF16)


def _xcorr(z, x, scale):
    """z, x: NHWC bf16 features -> fp32 [nx, 1, ho, wo]"""
    eng = shared_engine(x.device)
    nz, hz, wz, c = z.shape
    nx, h, w, c2 = x.shape
    assert c == c2 and nx % nz == 0, (z.shape, x.shape)
    out = torch.empty(nx, 1, h - hz + 1, w - wz + 1, device=x.device)
    eng.lib.xcorr_fwd(z, x, out, nz, nx, hz, wz, h, w, c, float(scale), eng.stream(x.device))
    return out


class SiamFC(nn.Module):
    def __init__(self, out_scale=0.001):
        super().__init__()
        self.out_scale = out_scale

    def forward(self, z, x):
        """z [nz,C,hz,wz], x [nx,C,h,w] (fp32 NCHW, as the backbone returns them) -> responses [nx,1,ho,wo]"""
        return _xcorr(_nhwc_bf16(z), _nhwc_bf16(x), self.out_scale)


class SiamConvFC(nn.Module):
    def __init__(self, in_channels, channels, num_convs=1, kernel_size=1, out_scale=0.001):
        super().__init__()
        if kernel_size != 1:
            raise NotImplementedError('SiamConvFC: the probe uses kernel_size=1 (siamfc_tracker_base.py:110-115)')
        self.out_scale = out_scale
        zc, xc, last = [], [], in_channels
        for _ in range(num_convs):
            zc.append(nn.Conv2d(last, channels, kernel_size))
            xc.append(nn.Conv2d(last, channels, kernel_size))
            last = channels
        self.z_convs, self.x_convs = nn.Sequential(*zc), nn.Sequential(*xc)
        self._packed = {}

    def _conv1x1(self, conv, t):
        """t NHWC bf16 -> conv(t) + bias, NHWC bf16 (vfs_conv_fwd; the bf16 copy of the weight is re-packed when it changes)"""
        eng = shared_engine(t.device)
        from .engine import params_epoch      # the probe's optimizers update weights through raw pointers (no _version bump)
        key = (conv.weight.data_ptr(), conv.weight._version, params_epoch())
        if self._packed.get(id(conv), (None,))[0] != key:
            wf = torch.empty(conv.out_channels, 1, 1, conv.in_channels, dtype=BF16, device=t.device)
            tab, n, total = build_pack_table([(conv.weight.data, wf, None, 0)], t.device)
            eng.lib.pack_weights(tab, n, total, eng.stream(t.device))
            self._packed[id(conv)] = (key, wf, tab)
        wf = self._packed[id(conv)][1]
        n, h, w, c = t.shape
        y = torch.empty(n, h, w, conv.out_channels, dtype=BF16, device=t.device)
        eng.lib.conv_fwd(t, wf, y, conv.bias.data if conv.bias is not None else None, None, n, h, w, c, h, w,
                         conv.out_channels, 1, 1, 1, 0, eng.stream(t.device))
        return y

    def forward(self, z, x):
        z, x = _nhwc_bf16(z), _nhwc_bf16(x)
        for cz, cx in zip(self.z_convs, self.x_convs):
            z, x = self._conv1x1(cz, z), self._conv1x1(cx, x)
        return _xcorr(z, x, self.out_scale)
