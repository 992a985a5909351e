"""Conv kernels (forward / dgrad / wgrad / stem) run through the CPU fiber emulator and compared
with torch conv2d on the same bf16-rounded operands.  CPU only."""
import pytest
import torch
import torch.nn.functional as F

from tests.emu_util import emu_lib, nchw, nhwc, rb, relerr
from vfs_amd.packing import build_pack_table, wgrad_splits


def pack(lib, w, stem=False):
    cout, cin, kh, kw = w.shape
    if stem:
        wf = torch.zeros(cout, 8, 8, 4, dtype=torch.bfloat16)
        wd = None
    else:
        wf = torch.empty(cout, kh, kw, cin, dtype=torch.bfloat16)
        wd = torch.empty(cin, kh, kw, cout, dtype=torch.bfloat16)
    tab, n, total = build_pack_table([(w, wf, wd, 1 if stem else 0)], 'cpu')
    lib.pack_weights(tab, n, total, None)
    return wf, wd


CASES = [  # N, H, W, Cin, Cout, k, stride, pad
    (2, 9, 11, 64, 64, 3, 1, 1),
    (1, 12, 10, 64, 128, 3, 2, 1),
    (3, 7, 7, 128, 64, 1, 1, 0),
    (2, 8, 8, 64, 128, 1, 2, 0),
]


@pytest.mark.parametrize('N,H,W,Cin,Cout,k,stride,pad', CASES)
def test_conv_fwd_dgrad_wgrad(N, H, W, Cin, Cout, k, stride, pad):
    lib = emu_lib()
    g = torch.Generator().manual_seed(N * 100 + H)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    wf, wd = pack(lib, w)
    assert torch.equal(wf.float(), w.permute(0, 2, 3, 1))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xh = nhwc(x)
    M = N * Ho * Wo
    y = torch.full((N, Ho, Wo, Cout), float('nan'), dtype=torch.bfloat16)
    nblk = (M + 127) // 128
    stats = torch.full((nblk, 2, Cout), float('nan'))
    bias = torch.randn(Cout, generator=g)
    lib.conv_fwd(xh, wf, y, bias, stats, N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, None)
    ref = F.conv2d(x, w, bias, stride, pad)
    assert relerr(nchw(y), ref) < 6e-3          # one bf16 rounding of the output
    yf = y.float().reshape(M, Cout)
    assert torch.isfinite(yf).all()
    for b in range(nblk):
        blk = yf[b * 128:(b + 1) * 128]
        assert torch.allclose(stats[b, 0], blk.sum(0), rtol=1e-4, atol=1e-3)
        assert torch.allclose(stats[b, 1], (blk * blk).sum(0), rtol=1e-4, atol=1e-3)

    # ---- dgrad (+ fused residual-gradient add) and wgrad vs autograd
    dy = rb(torch.randn(N, Cout, Ho, Wo, generator=g))
    add = rb(torch.randn(N, Cin, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride, pad).backward(dy)
    dx = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16)
    lib.conv_dgrad(nhwc(dy), wd, dx, nhwc(add), N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, None)
    assert relerr(nchw(dx), xr.grad + add) < 6e-3
    nsplit, pps = wgrad_splits(M, Cout, k * k * Cin, target_blocks=12)
    partial = torch.full((nsplit, Cout, k * k * Cin), float('nan'))
    grad = torch.ones(Cout, Cin, k, k)
    lib.conv_wgrad(nhwc(dy), xh, partial, grad, N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, nsplit, pps, None)
    assert relerr(grad - 1.0, wr.grad) < 2e-4    # fp32 accumulate / fp32 output, accumulates into grad


def test_stem_fwd_wgrad():
    lib = emu_lib()
    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 20, 18
    x = rb(torch.randn(N, 3, H, W, generator=g))
    w = rb(torch.randn(64, 3, 7, 7, generator=g) * 0.1)
    wf, _ = pack(lib, w, stem=True)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    # reference input layout [B][V][3][T][H][W] with B=N, V=1, T=1
    x4 = torch.full((N, H, W, 4), float('nan'), dtype=torch.bfloat16)
    lib.imgs_to_nhwc4(x.reshape(N, 1, 3, 1, H, W).contiguous(), x4, N, 1, 1, H, W, W, None)
    assert torch.equal(x4[..., :3].float(), x.permute(0, 2, 3, 1)) and (x4[..., 3] == 0).all()
    M = N * Ho * Wo
    y = torch.full((N, Ho, Wo, 64), float('nan'), dtype=torch.bfloat16)
    stats = torch.zeros((M + 127) // 128, 2, 64)
    lib.stem_fwd(x4, wf, y, stats, N, H, W, Ho, Wo, None)
    ref = F.conv2d(x, w, None, 2, 3)
    assert relerr(nchw(y), ref) < 6e-3
    dy = rb(torch.randn(N, 64, Ho, Wo, generator=g))
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, 2, 3).backward(dy)
    nsplit, pps = wgrad_splits(M, 64, 256, target_blocks=6)
    partial = torch.zeros(nsplit, 64, 256)
    grad = torch.zeros(64, 3, 7, 7)
    lib.stem_wgrad(nhwc(dy), x4, partial, grad, N, H, W, Ho, Wo, nsplit, pps, None)
    assert relerr(grad, wr.grad) < 2e-4
