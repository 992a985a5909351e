"""CosineSimLoss under the reference's registry name and constructor
(mmaction/models/losses/sim_loss.py:25-63, losses/base.py:6-37).  The arithmetic
(L2-normalise, row dot product, 2-2*cos or -cos) is the fused HIP kernel
vfs_cosine_loss_fwd/bwd, evaluated for all temporal rolls of the head loss at once."""
import torch
import torch.nn as nn

from .engine import BF16, shared_engine
from .registry import LOSSES


class _CosineLossFn(torch.autograd.Function):
    """loss[i] = L(p[i], z[i]) for [N,C] inputs; gradient flows to p only when z is detached."""

    @staticmethod
    def forward(ctx, p, z, negative):
        eng = shared_engine()
        N, C = p.shape
        pb, zb = p.detach().to(BF16).contiguous(), z.detach().to(BF16).contiguous()
        loss = torch.empty(1, N, dtype=torch.float32, device=p.device)
        # the kernel computes 0.5*L(p1,z2)+0.5*L(p2,z1); feeding (p,z,p,z) yields L(p,z)
        eng.lib.cosine_loss_fwd(pb, zb, pb, zb, loss, N, C, 1, 1, int(negative), 1.0, eng.stream(p.device))
        ctx.save_for_backward(pb, zb)
        ctx.negative = negative
        ctx.z_needs = z.requires_grad
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        pb, zb = ctx.saved_tensors
        eng = shared_engine()
        N, C = pb.shape
        dev = pb.device
        gl = g.reshape(1, N).contiguous().float()
        dp = torch.empty(N, C, dtype=BF16, device=dev)
        scratch = torch.empty(N, C, dtype=BF16, device=dev)
        s = eng.stream(dev)
        # view 0 pairs p1 with z2, view 1 pairs p2 with z1; weight 2.0 undoes the kernel's 0.5
        eng.lib.cosine_loss_bwd(pb, zb, pb, zb, gl, dp, scratch, N, C, 1, 1, int(ctx.negative), 2.0, s)
        dz = None
        if ctx.z_needs:
            dzb = torch.empty(N, C, dtype=BF16, device=dev)
            eng.lib.cosine_loss_bwd(zb, pb, zb, pb, gl, dzb, scratch, N, C, 1, 1, int(ctx.negative), 2.0, s)
            dz = dzb.float()
        return dp.float(), dz, None


@LOSSES.register_module()
class CosineSimLoss(nn.Module):
    def __init__(self, with_norm=True, negative=False, pairwise=False, loss_weight=1.0, **kwargs):
        super().__init__()
        if not with_norm or pairwise:
            raise NotImplementedError('HIP path covers with_norm=True, pairwise=False (the shipped configs)')
        self.with_norm, self.negative, self.pairwise, self.loss_weight = with_norm, negative, pairwise, loss_weight

    def forward(self, cls_score, label, mask=None, **kwargs):
        assert mask is None
        assert cls_score.ndim == 2 and cls_score.shape == label.shape
        return _CosineLossFn.apply(cls_score, label, self.negative) * self.loss_weight
