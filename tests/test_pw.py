"""The persistent producer / consumer 1x1 kernel (csrc/conv_pw.hip) against the one-tile-per-workgroup implicit-GEMM kernel
(csrc/conv_igemm.hip) and against torch: same MFMA order per tile and the same epilogue, so every output - activations,
statistics rows, fused BatchNorm-backward rows - must be BIT-identical between the two; torch conv2d (fp32, same bf16
operands) bounds both.  backend=emu: host build through the fiber emulator; backend=gpu: libvfs_hip.so on the MI355X."""
import pytest
import torch
import torch.nn.functional as F

from tests.emu_util import nchw, nhwc, pack_relu_mask, rb, relerr
from tests.test_emu_conv import pack

SHAPES = [  # N, H, W, Cin, Cout
    (5, 16, 16, 128, 256),    # 20 tiles of 128 x 128, two K-steps: several tiles per workgroup, ring wraps inside and across tiles
    (20, 8, 8, 64, 256),      # one K-step per tile: every step is a tile's last
    (3, 7, 7, 512, 64),       # 64-channel tiles (five-stage ring), eight K-steps, ragged M = 147
    (2, 12, 12, 320, 72),     # ragged channel tile (72 = 64 + 8), odd number of K-steps
    (4, 16, 16, 1024, 256),   # sixteen K-steps (ResNet-50 layer3 conv1 / conv3 dgrad), two tiles per workgroup
    (1, 9, 14, 64, 128),      # fewer tiles than workgroups would be: one tile, ragged M = 126
]


def both(lib, fn):
    """run fn() with the persistent kernel forced (igemm_pw = 2) and switched off (0); returns (pw, plain).  The plain kernel
    runs with the same channel tile (128 wherever Cout % 128 == 0: igemm_narrow_below = 0) - the fused statistics rows are fp32
    sums whose grouping follows the tile"""
    outs = []
    lib.set_option(b'igemm_narrow_below', 0)
    lib.set_option(b'igemm_ring_mfma32', 0)      # (the plain kernel's ring on 16x16x32 MFMAs, as the persistent kernel's K-steps)
    for flag in (2, 0):
        lib.set_option(b'igemm_pw', flag)
        try:
            outs.append(fn())
        finally:
            lib.set_option(b'igemm_pw', 0)
    lib.set_option(b'igemm_narrow_below', 513)
    lib.set_option(b'igemm_ring_mfma32', 1)
    return outs


@pytest.mark.parametrize('N,H,W,Cin,Cout', SHAPES)
def test_pw_forward_and_dgrad(backend, N, H, W, Cin, Cout):
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(N * 17 + Cin + Cout)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5)
    wf, wd = pack(backend, w)
    M, nblk = N * H * W, (N * H * W + 127) // 128
    bias = torch.randn(Cout, generator=g)
    dy = rb(torch.randn(N, Cout, H, W, generator=g))
    add = rb(torch.randn(N, Cin, H, W, generator=g))
    xh, dyh, addh = d(nhwc(x)), d(nhwc(dy)), d(nhwc(add))

    def fwd(b):
        y = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
        stats = torch.full((nblk, 2, Cout), float('nan'), device=dev)
        lib.conv_fwd(xh, wf, y, d(bias) if b else None, stats, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, None)
        return y.cpu(), stats.cpu()

    def dgrad(with_add):
        dx = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
        lib.conv_dgrad(dyh, wd, dx, addh if with_add else None, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, None)
        return dx.cpu()

    for b in (False, True):       # bias-free: statistics rows on the matrix cores; with bias: per-element
        (y1, s1), (y0, s0) = both(lib, lambda: fwd(b))
        assert torch.equal(y1, y0) and torch.equal(s1, s0)
        assert relerr(nchw(y1), F.conv2d(x, w, bias if b else None)) < 6e-3
        yf = y1.float().reshape(M, Cout).double()
        assert torch.allclose(s1[:, 0].double().sum(0), yf.sum(0), rtol=1e-4, atol=5e-3)
        assert torch.allclose(s1[:, 1].double().sum(0), (yf * yf).sum(0), rtol=1e-4, atol=5e-3)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w).backward(dy)
    for with_add in ((False, True) if Cout % 64 == 0 else ()):      # the dgrad's K is Cout
        dx1, dx0 = both(lib, lambda: dgrad(with_add))
        assert torch.equal(dx1, dx0)
        assert relerr(nchw(dx1), xr.grad + (add if with_add else 0)) < 6e-3


@pytest.mark.parametrize('N,H,W,Cin,Cout,G', [
    (4, 16, 16, 128, 256, 2),     # dgrad producing 128 channels (one 128-channel tile), two statistics groups, four K-steps
    (6, 8, 8, 64, 512, 1),        # 64-channel tile, eight K-steps, three pixel blocks
    (3, 7, 7, 256, 128, 1),       # ragged M = 147, two channel tiles
])
@pytest.mark.parametrize('mask', ['bits', 'relu', 'none'])
def test_pw_dgrad_fused_epilogues(backend, N, H, W, Cin, Cout, G, mask):
    """the dgrad epilogues of the train step on the persistent kernel: + identity gradient gated by the bit-packed mask,
    + fused BatchNorm-backward statistics rows - bit-identical to the plain kernel (whose rows tests/test_emu_conv.py holds
    against bn_bwd_reduce's definition)"""
    lib, d, dev = backend.lib, backend.d, backend.dev
    gen = torch.Generator().manual_seed(N * 5 + Cin + Cout)
    w = rb(torch.randn(Cout, Cin, 1, 1, generator=gen) * (2.0 / Cin) ** 0.5)
    _, wd = pack(backend, w)
    M = N * H * W
    mpg, nblk = M // G, (M + 127) // 128
    if G > 1 and mpg % 128:
        pytest.skip('blocks must not straddle groups')
    dy = d(nhwc(rb(torch.randn(N, Cout, H, W, generator=gen))))
    gid = rb(torch.randn(N, H, W, Cin, generator=gen))
    ypre = rb(torch.relu(torch.randn(N, H, W, Cin, generator=gen)))
    bits = d(pack_relu_mask(ypre))
    x = rb(torch.randn(N, H, W, Cin, generator=gen) * 1.5 + 0.3)
    bnp = torch.stack([torch.rand(G, Cin, generator=gen) + 0.5, torch.randn(G, Cin, generator=gen), torch.randn(G, Cin, generator=gen) * 0.1,
                       torch.rand(G, Cin, generator=gen) + 0.5], 1).contiguous()
    ymask = d(pack_relu_mask(rb(torch.relu(torch.randn(N, H, W, Cin, generator=gen))))) if mask == 'bits' else None
    relu = {'relu': 1, 'bits': 2}.get(mask, 0)

    def run():
        dx = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
        part = torch.full((nblk, 2, Cin), float('nan'), device=dev)
        lib.conv_dgrad_bn_maskadd(dy, wd, dx, d(gid.to(torch.bfloat16)), bits, d(x.to(torch.bfloat16)), ymask, d(bnp), part, mpg, relu,
                                  N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, None)
        dx2 = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
        lib.conv_dgrad_maskadd(dy, wd, dx2, d(gid.to(torch.bfloat16)), bits, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, None)
        return dx.cpu(), part.cpu(), dx2.cpu()
    (dx1, p1, dm1), (dx0, p0, dm0) = both(lib, run)
    assert torch.isfinite(p1).all() and torch.isfinite(dx1.float()).all()
    assert torch.equal(dx1, dx0) and torch.equal(p1, p0) and torch.equal(dm1, dm0) and torch.equal(dx1, dm1)


@pytest.mark.gpu
@pytest.mark.parametrize('N,H,W,Cin,Cout', [
    (64, 64, 64, 64, 256), (64, 64, 64, 256, 64), (64, 32, 32, 128, 512), (64, 32, 32, 512, 128),
    (64, 16, 16, 256, 1024), (64, 16, 16, 1024, 256), (64, 8, 8, 512, 2048), (64, 8, 8, 2048, 512),
])
def test_pw_bench_shapes_bit_identical(gpu_backend, N, H, W, Cin, Cout):
    """every 1x1 layer of ResNet-50 at the bench batch (64 frames of 256 x 256): forward (+ statistics rows) and dgrad of the
    persistent kernel equal the plain kernel bit for bit - 16 tiles per workgroup on the large maps, real asynchrony"""
    lib, dev = gpu_backend.lib, gpu_backend.dev
    g = torch.Generator(device=dev).manual_seed(Cin + Cout)
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).to(torch.bfloat16)
    wf = (torch.randn(Cout, 1, 1, Cin, device=dev, generator=g) * (2.0 / Cin) ** 0.5).to(torch.bfloat16)
    wd = (torch.randn(Cin, 1, 1, Cout, device=dev, generator=g) * (2.0 / Cout) ** 0.5).to(torch.bfloat16)
    dy = torch.randn(N, H, W, Cout, device=dev, generator=g).to(torch.bfloat16)
    M = N * H * W
    s = torch.cuda.current_stream().cuda_stream

    def run():
        y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
        st = torch.empty((M + 127) // 128, 2, Cout, device=dev)
        dx = torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
        for _ in range(3):      # repeated: a hand-off race would not repeat
            lib.conv_fwd(x, wf, y, None, st, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, s)
            lib.conv_dgrad(dy, wd, dx, None, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, s)
        torch.cuda.synchronize()
        return y.clone(), st.clone(), dx.clone()
    (y1, s1, d1), (y0, s0, d0) = both(lib, run)
    assert torch.equal(y1, y0) and torch.equal(s1, s0) and torch.equal(d1, d0)
    ref = torch.nn.functional.conv2d(x[:2].float().permute(0, 3, 1, 2), wf.float().permute(0, 3, 1, 2))
    assert relerr(y1[:2].float().permute(0, 3, 1, 2), ref) < 6e-3


@pytest.mark.parametrize('M,Cin,Cout', [(64, 2048, 512), (40, 256, 64), (128, 512, 2048), (8, 128, 16), (97, 384, 48)])
def test_skinny_linear_layers(backend, M, Cin, Cout):
    """the skinny GEMM of the head's Linear layers (at most 128 rows: four K quarters per workgroup, fragments straight from
    global memory) - forward with bias, dgrad with a residual operand - against torch (fp32, same bf16 operands) and against
    the implicit-GEMM kernel (equal up to one bf16 rounding: the K quarters are summed in another order)"""
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(M + Cin + Cout)
    x = rb(torch.randn(M, Cin, generator=g))
    w = rb(torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5)
    bias = torch.randn(Cout, generator=g)
    wf, wd = pack(backend, w.reshape(Cout, Cin, 1, 1))
    outs = []
    for flag in (1, 0):
        lib.set_option(b'igemm_skinny', flag)
        try:
            y = torch.full((M, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
            lib.conv_fwd(d(x.to(torch.bfloat16)), wf, y, d(bias), None, M, 1, 1, Cin, 1, 1, Cout, 1, 1, 1, 0, None)
            res = [y.float().cpu()]
            if Cout % 128 == 0 and Cin % 16 == 0:     # the dgrad's K is Cout
                dy = rb(torch.randn(M, Cout, generator=torch.Generator().manual_seed(7)))
                add = rb(torch.randn(M, Cin, generator=torch.Generator().manual_seed(8)))
                dx = torch.full((M, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
                lib.conv_dgrad(d(dy.to(torch.bfloat16)), wd, dx, d(add.to(torch.bfloat16)), M, 1, 1, Cin, 1, 1, Cout, 1, 1, 1, 0, None)
                res.append(dx.float().cpu())
                assert relerr(res[1], dy @ w + add) < 6e-3
            outs.append(res)
        finally:
            lib.set_option(b'igemm_skinny', 1)
    assert relerr(outs[0][0], x @ w.t() + bias) < 6e-3
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.isfinite(a_).all() and relerr(a_, b_) < 8e-3
