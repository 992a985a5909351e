"""CosineSimLoss under the reference's registry name and constructor
(mmaction/models/losses/sim_loss.py:25-63, losses/base.py:6-37).

Two execution paths, both HIP:
  * [N,C] operands with L2 normalisation (the shipped configs, sim_siam_head.py:165-174): the fused bf16 kernel
    vfs_cosine_loss_fwd/bwd, which the train step evaluates for all temporal rolls of the head loss at once;
  * everything else the reference class accepts - spatial operands [B,C,*], `pairwise=True` (the affinity matrix
    einsum('bci,bcj->bij') with an optional mask, :48-56), `with_norm=False` - on the fp32 kernels of csrc/simloss.hip:
    the affinity is a dense contraction and runs on the matrix cores (v_mfma_f32_32x32x2_f32), forward and backward."""
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


class _SpatialSimLossFn(torch.autograd.Function):
    """sim_loss.py:42-63 on [B,C,S] fp32 operands (S = flattened positions; [N,C] is S = 1): optional F.normalize over C, then
    pairwise=True: mean over (i, j) of einsum('bci,bcj->bij') [* mask]; pairwise=False: mean over positions of sum_c a*l."""

    @staticmethod
    def forward(ctx, a, l, mask, with_norm, negative, pairwise):
        eng = shared_engine()
        lib, dev = eng.lib, a.device
        s = eng.stream(dev)
        B, C = a.shape[:2]
        a3 = a.detach().float().reshape(B, C, -1).contiguous()
        l3 = l.detach().float().reshape(B, C, -1).contiguous()
        Sa, Sl = a3.shape[2], l3.shape[2]
        m3 = None
        if mask is not None:
            assert pairwise, 'a mask needs pairwise=True (sim_loss.py:46-47)'
            assert tuple(mask.shape) == (B, Sa, Sl), (tuple(mask.shape), (B, Sa, Sl))      # sim_loss.py:53
            m3 = mask.detach().float().contiguous()
        inva = invl = None
        if with_norm:
            inva = torch.empty(B, Sa, dtype=torch.float32, device=dev)
            invl = torch.empty(B, Sl, dtype=torch.float32, device=dev)
            lib.simloss_colnorm(a3, inva, B, C, Sa, s)
            lib.simloss_colnorm(l3, invl, B, C, Sl, s)
        tiles = ((Sa + 31) // 32) * (((Sl + 31) // 32) if pairwise else 1)      # diagonal mode: the diagonal tiles only
        partial = torch.empty(B * tiles, dtype=torch.float32, device=dev)
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        lib.simloss_fwd(a3, l3, inva, invl, m3, partial, loss, B, C, Sa, Sl, int(pairwise), int(negative), 1.0, s)
        ctx.save_for_backward(a3, l3, inva, invl, m3)
        ctx.cfg = (negative, pairwise, a.shape, l.shape, a.dtype, l.dtype)
        ctx.needs = (a.requires_grad, l.requires_grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        a3, l3, inva, invl, m3 = ctx.saved_tensors
        negative, pairwise, ashape, lshape, adt, ldt = ctx.cfg
        eng = shared_engine()
        lib, dev = eng.lib, a3.device
        s = eng.stream(dev)
        B, C, Sa = a3.shape
        Sl = l3.shape[2]
        gl = g.detach().float().contiguous()

        def side(x, invx, other, invo, Sx, So, transposed):
            d = torch.empty(B, C, Sx, dtype=torch.float32, device=dev)
            lib.simloss_bwd(other, invo, m3, int(transposed), gl, d, B, C, Sx, So, int(pairwise), int(negative), 1.0, s)
            if invx is None:
                return d
            dx = torch.empty_like(d)
            lib.simloss_norm_bwd(x, invx, d, dx, B, C, Sx, s)
            return dx
        da = side(a3, inva, l3, invl, Sa, Sl, False).reshape(ashape).to(adt) if ctx.needs[0] else None
        dl = side(l3, invl, a3, inva, Sl, Sa, True).reshape(lshape).to(ldt) if ctx.needs[1] else None
        return da, dl, None, None, None, None


@LOSSES.register_module()
class CosineSimLoss(nn.Module):
    def __init__(self, with_norm=True, negative=False, pairwise=False, loss_weight=1.0, **kwargs):
        super().__init__()
        self.with_norm, self.negative, self.pairwise, self.loss_weight = with_norm, negative, pairwise, loss_weight

    def forward(self, cls_score, label, mask=None, **kwargs):
        if mask is not None:
            assert self.pairwise                                           # sim_loss.py:46-47
        if self.pairwise and cls_score.ndim < 3:
            raise IndexError('pairwise=True flattens from dim 2: the operands must be [B,C,*] (sim_loss.py:49-50)')
        if cls_score.ndim == 2 and self.with_norm and not self.pairwise:
            assert cls_score.shape == label.shape
            return _CosineLossFn.apply(cls_score, label, self.negative) * self.loss_weight
        assert cls_score.shape[:2] == label.shape[:2]
        if not self.pairwise:
            assert cls_score.shape == label.shape
        return _SpatialSimLossFn.apply(cls_score, label, mask, self.with_norm, self.negative, self.pairwise) * self.loss_weight
