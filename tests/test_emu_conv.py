"""Conv kernels (forward / dgrad / wgrad / stem) against torch conv2d (CPU, fp32) on the same
bf16-rounded operands.  backend=emu: host build through the fiber emulator (CPU);
backend=gpu: libvfs_hip.so on the MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.emu_util import pack_relu_mask, nchw, nhwc, rb, relerr
from vfs_amd.packing import build_pack_table, conv_halo_eligible, conv_stats_rows, wgrad_halo_eligible, wgrad_inl_floats, wgrad_splits


def pack(be, w, stem=False):
    cout, cin, kh, kw = w.shape
    dev = be.dev
    w = w.to(dev)
    if stem:
        wf = torch.zeros(cout, 8, 8, 4, dtype=torch.bfloat16, device=dev)
        wd = None
    else:
        wf = torch.empty(cout, kh, kw, cin, dtype=torch.bfloat16, device=dev)
        wd = torch.empty(cin, kh, kw, cout, dtype=torch.bfloat16, device=dev)
    tab, n, total = build_pack_table([(w, wf, wd, 1 if stem else 0)], dev)
    be.lib.pack_weights(tab, n, total, None)
    return wf, wd


def run_conv_case(be, N, H, W, Cin, Cout, k, stride, pad, wgrad_blocks=12):
    lib, d = be.lib, be.d
    g = torch.Generator().manual_seed(N * 100 + H)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    wf, wd = pack(be, w)
    assert torch.equal(wf.float().cpu(), w.permute(0, 2, 3, 1))
    assert torch.equal(wd.float().cpu(), w.permute(1, 2, 3, 0))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xh = d(nhwc(x))
    M = N * Ho * Wo
    y = torch.full((N, Ho, Wo, Cout), float('nan'), dtype=torch.bfloat16, device=be.dev)
    nblk = conv_stats_rows(N, 1, H, W, Cin, Cout, k, stride, pad, Ho, Wo)     # spatial tiles (halo kernels) or linear blocks
    stats = torch.full((nblk, 2, Cout), float('nan'), device=be.dev)
    bias = torch.randn(Cout, generator=g)
    lib.conv_fwd(xh, wf, y, d(bias), stats, N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, None)
    ref = F.conv2d(x, w, bias, stride, pad)
    assert relerr(nchw(y.cpu()), ref) < 6e-3          # one bf16 rounding of the output
    yf = y.float().cpu().reshape(M, Cout)
    assert torch.isfinite(yf).all()
    # BatchNorm partials: one (sum, sumsq) row per 128-pixel block (linear blocks in the generic
    # kernel, spatial tiles in the halo kernel); what the consumers rely on is that the blocks of
    # the first / second half of the batch (the two views) are the first / second half of the rows
    st = stats.cpu()
    halves = 2 if (N % 2 == 0 and conv_stats_rows(N, 2, H, W, Cin, Cout, k, stride, pad, Ho, Wo) is not None) else 1
    for h in range(halves):
        rows = slice(h * nblk // halves, (h + 1) * nblk // halves)
        pix = yf[h * M // halves:(h + 1) * M // halves].double()
        assert torch.allclose(st[rows, 0].double().sum(0), pix.sum(0), rtol=1e-4, atol=5e-3)
        assert torch.allclose(st[rows, 1].double().sum(0), (pix * pix).sum(0), rtol=1e-4, atol=5e-3)

    # ---- dgrad (+ fused residual-gradient add) and wgrad vs autograd
    dy = rb(torch.randn(N, Cout, Ho, Wo, generator=g))
    add = rb(torch.randn(N, Cin, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride, pad).backward(dy)
    dx = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=be.dev)
    lib.conv_dgrad(d(nhwc(dy)), wd, dx, d(nhwc(add)), N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, None)
    assert relerr(nchw(dx.cpu()), xr.grad + add) < 6e-3
    halo = (N, H, W, Cin) if wgrad_halo_eligible(N, H, W, Cin, Cout, k, stride, pad) else None
    nsplit, pps = wgrad_splits(M, Cout, k * k * Cin, target_blocks=wgrad_blocks, halo_geom=halo)
    partial = torch.full((nsplit, Cout, k * k * Cin), float('nan'), device=be.dev)
    grad = torch.ones(Cout, Cin, k, k, device=be.dev)
    lib.conv_wgrad(d(nhwc(dy)), xh, partial, grad, N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, nsplit, pps, None)
    assert relerr(grad.cpu() - 1.0, wr.grad) < 3e-4    # fp32 accumulate / fp32 output, accumulates into grad
    if Cin % 4 == 0:
        # round 6: the same gradient with the split-K reduction inside the launch (vfs_conv_wgrad_inl: the last workgroup of a tile
        # sums the partials in split order): equal to the two-launch form up to the fp32 order of the splits, run-to-run
        # bit-identical, tickets left at zero
        tickets = torch.zeros(lib.cfunc('wgrad_tickets')(), dtype=torch.int32, device=be.dev)
        gi = [torch.ones(Cout, Cin, k, k, device=be.dev) for _ in range(2)]
        wsi = torch.empty(wgrad_inl_floats(nsplit, Cout, k * k * Cin), device=be.dev)
        for gq in gi:
            wsi.fill_(float('nan'))
            lib.conv_wgrad_inl(d(nhwc(dy)), xh, None, 0, wsi, gq, tickets, N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, nsplit, pps, None)
        assert torch.equal(gi[0].cpu(), gi[1].cpu())
        assert int(tickets.cpu().abs().sum()) == 0
        assert relerr(gi[0].cpu(), grad.cpu()) < 2e-6


CASES = [  # N, H, W, Cin, Cout, k, stride, pad
    (2, 9, 11, 64, 64, 3, 1, 1),
    (1, 12, 10, 64, 128, 3, 2, 1),
    (3, 7, 7, 128, 64, 1, 1, 0),
    (2, 8, 8, 64, 128, 1, 2, 0),
    (2, 8, 8, 64, 64, 1, 1, 0),       # one K-step (Ktot == 64): single-buffer variant, 64-channel tile
    (1, 12, 12, 64, 256, 1, 1, 0),    # one K-step, two 128-channel tiles, ragged M = 144
    (3, 7, 7, 512, 64, 1, 1, 0),      # DMA-ring variant (1x1, K >= 256): 8 K-steps, 64-channel tile, ragged M = 147
    (2, 8, 8, 256, 256, 1, 1, 0),     # DMA-ring variant: 4 K-steps forward and dgrad, two 128-channel tiles
    (1, 5, 5, 320, 128, 1, 1, 0),     # DMA-ring variant: odd number of K-steps (5), one ragged tile
    (2, 9, 11, 64, 64, 3, 2, 1),      # odd sizes: unequal parity classes in the stride-2 dgrad
    (1, 16, 32, 128, 64, 3, 1, 1),    # halo-tile kernel: 8x16 spatial tiles, two channel chunks
    (4, 8, 8, 64, 128, 3, 1, 1),      # halo-tile kernel: two whole 8x8 images per workgroup
    (2, 32, 32, 64, 64, 3, 1, 1),     # halo-tile kernel, 64 output channels: 16x16 tiles, 4 pixel waves
    (2, 14, 14, 64, 128, 3, 1, 1),    # halo-tile kernel, RAGGED 8x16 tiles (14 x 14 map of a 224 x 224 input)
    (1, 28, 28, 64, 64, 3, 1, 1),     # halo-tile kernel, RAGGED 16x16 tiles (28 x 28)
    (3, 7, 14, 128, 128, 3, 1, 1),    # halo-tile kernel, ragged bottom row only, odd image count
    (4, 7, 7, 64, 128, 3, 1, 1),      # halo-tile kernel, whole 7x7 images in pairs (8x8 tiles, masked)
    (2, 16, 16, 64, 128, 3, 2, 1),    # stride-2 3x3 on an evenly tiled map: linear-address weight gradient with per-step row validity (LIN = 2)
    (3, 32, 32, 128, 64, 3, 2, 1),    # the same, 16-wide output rows, odd image count, two K-steps per tap
    (4, 8, 8, 128, 128, 1, 2, 0),     # stride-2 1x1 (downsample unit), 4-wide output rows: sixteen rows per 64-pixel step
]


@pytest.mark.parametrize('N,H,W,Cin,Cout,k,stride', [
    (3, 7, 7, 512, 64, 1, 1),       # DMA ring, ragged M = 147 (rows past M are computed from a re-fetched row)
    (2, 2, 2, 1024, 256, 1, 1),     # DMA ring, 8 rows only (ResNet-50 layer4 on a 64 x 64 input)
    (1, 12, 12, 64, 256, 1, 1),     # one K-step, two 128-channel tiles, ragged M = 144
    (3, 7, 7, 128, 72, 1, 1),       # ragged channel tile (72 = 64 + 8)
    (2, 9, 11, 64, 64, 3, 2),       # 3x3 stride 2 through the implicit-GEMM kernel
])
def test_forward_statistics_rows_without_bias(backend, N, H, W, Cin, Cout, k, stride):
    """the BatchNorm statistics rows of a bias-free forward conv (the ConvModule case): the implicit-GEMM kernel computes them
    on the matrix cores from the staged bf16 tile (option igemm_mfma_stats) - equal to the sums over the stored output"""
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(N * 31 + Cin + Cout)
    pad = k // 2
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    wf, _ = pack(backend, w)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    M = N * Ho * Wo
    nblk = conv_stats_rows(N, 1, H, W, Cin, Cout, k, stride, pad, Ho, Wo)
    outs = []
    for flag in (1, 0):
        lib.set_option(b'igemm_mfma_stats', flag)
        try:
            y = torch.full((N, Ho, Wo, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
            stats = torch.full((nblk, 2, Cout), float('nan'), device=dev)
            lib.conv_fwd(d(nhwc(x)), wf, y, None, stats, N, H, W, Cin, Ho, Wo, Cout, k, k, stride, pad, None)
        finally:
            lib.set_option(b'igemm_mfma_stats', 1)
        yf = y.float().cpu().reshape(M, Cout).double()
        st = stats.cpu().double()
        assert torch.isfinite(st).all()
        assert torch.allclose(st[:, 0].sum(0), yf.sum(0), rtol=1e-5, atol=2e-3)
        assert torch.allclose(st[:, 1].sum(0), (yf * yf).sum(0), rtol=1e-5, atol=2e-3)
        outs.append((y.cpu(), st))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize('M_img,nsplit,pps', [(8, 1, 64), (16, 1, 128), (24, 1, 192), (40, 1, 320), (40, 2, 192), (72, 3, 192), (33, 1, 320)])
def test_wgrad_pixel_step_counts(backend, M_img, nsplit, pps):
    """the generic weight-gradient kernel loads pixel steps in PAIRS (two register stages): 1, 2, 3 (odd) and 5 steps per
    split, splits whose last steps run past M, against autograd"""
    lib, d, dev = backend.lib, backend.d, backend.dev
    N, H, W, Cin, Cout = 1, M_img, 8, 128, 64
    M = N * H * W
    assert nsplit * pps >= M
    g = torch.Generator().manual_seed(M + nsplit)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1)
    dy = rb(torch.randn(N, Cout, H, W, generator=g))
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, 1, 0).backward(dy)
    partial = torch.full((nsplit, Cout, Cin), float('nan'), device=dev)
    grad = torch.zeros(Cout, Cin, 1, 1, device=dev)
    lib.conv_wgrad(d(nhwc(dy)), d(nhwc(x)), partial, grad, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, nsplit, pps, None)
    assert relerr(grad.cpu(), wr.grad) < 3e-4


@pytest.mark.parametrize('M_img,Cin,Cout,nsplit,pps', [(8, 128, 128, 1, 64), (16, 128, 128, 1, 128), (24, 256, 128, 1, 192), (40, 128, 256, 1, 320),
                                                    (40, 128, 128, 2, 192), (72, 256, 256, 3, 192), (33, 128, 128, 1, 320), (64, 128, 128, 1, 512),
                                                    (50, 256, 128, 4, 128)])
def test_wgrad_ring_matches_staged_kernel(backend, M_img, Cin, Cout, nsplit, pps):
    """the LDS-DMA ring of the 1x1 weight gradient (conv_wgrad_ring_kernel: unpadded, XOR-swizzled stages, 1-8 steps per split,
    splits and steps that run past M, several k-column / cout tiles) against the register-staged kernel: bit-identical partials
    and gradients; and against autograd"""
    lib, d, dev = backend.lib, backend.d, backend.dev
    N, H, W = 1, M_img, 8
    M = N * H * W
    assert nsplit * pps >= M
    g = torch.Generator().manual_seed(M + nsplit + Cin)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1)
    dy = rb(torch.randn(N, Cout, H, W, generator=g))
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, 1, 0).backward(dy)
    outs = []
    try:
        for ring in (1, 0):
            lib.set_option(b'wgrad_ring', ring)
            partial = torch.full((nsplit, Cout, Cin), float('nan'), device=dev)
            grad = torch.zeros(Cout, Cin, 1, 1, device=dev)
            lib.conv_wgrad(d(nhwc(dy)), d(nhwc(x)), partial, grad, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, nsplit, pps, None)
            outs.append((partial.cpu(), grad.cpu()))
    finally:
        lib.set_option(b'wgrad_ring', 1)
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    assert relerr(outs[0][1], wr.grad) < 3e-4


@pytest.mark.parametrize('N,H,W,Cin,Cout,k,stride,pad', CASES)
def test_conv_fwd_dgrad_wgrad(backend, N, H, W, Cin, Cout, k, stride, pad):
    run_conv_case(backend, N, H, W, Cin, Cout, k, stride, pad)


@pytest.mark.parametrize('N,H,W,Cin,Cout,k,stride,pad', [(3, 7, 7, 128, 64, 1, 1, 0), (2, 9, 11, 64, 64, 3, 2, 1), (1, 12, 12, 256, 128, 1, 1, 0)])
@pytest.mark.parametrize('onek', [3, 0])
def test_conv_single_buffer_variant(backend, N, H, W, Cin, Cout, k, stride, pad, onek):
    """the single-buffer implicit-GEMM kernel on multi-K-step problems of every kind (option igemm_onek = 3, the default since
    round 6) and the double-buffer pipeline it replaced (igemm_onek = 0), without the DMA ring taking the 1x1 cases"""
    backend.lib.set_option(b'igemm_onek', onek)
    backend.lib.set_option(b'igemm_ring_tiles', 0)
    try:
        run_conv_case(backend, N, H, W, Cin, Cout, k, stride, pad)
    finally:
        backend.lib.set_option(b'igemm_onek', 3)
        backend.lib.set_option(b'igemm_ring_tiles', 512)


@pytest.mark.parametrize('shape', [(48, 40, 3, 3), (70, 33, 1, 1), (5, 3, 5, 5), (64, 64, 3, 3), (130, 257, 1, 1),
                                   (128, 192, 1, 1), (64, 128, 3, 3), (96, 64, 3, 3), (192, 64, 1, 1)])
def test_pack_weights_ragged(backend, shape):
    """tiled LDS transpose of the weight repack on tile-ragged / odd shapes, several tensors per launch; the last four shapes (and
    the second tensor of (64, 64, 3, 3)) take the 16-byte tile variants (kinds 2 / 3 of the pack table)"""
    g = torch.Generator().manual_seed(shape[0])
    dev = backend.dev
    ws = [torch.randn(*shape, generator=g), torch.randn(shape[1], shape[0], 1, 1, generator=g)]
    entries = []
    for w in ws:
        co, ci, kh, kw = w.shape
        entries.append((w.to(dev), torch.full((co, kh, kw, ci), float('nan'), dtype=torch.bfloat16, device=dev),
                        torch.full((ci, kh, kw, co), float('nan'), dtype=torch.bfloat16, device=dev), 0))
    tab, n, total = build_pack_table(entries, dev)
    backend.lib.pack_weights(tab, n, total, None)
    for (w, wf, wd, _), w0 in zip(entries, ws):
        assert torch.equal(wf.float().cpu(), rb(w0).permute(0, 2, 3, 1))
        assert torch.equal(wd.float().cpu(), rb(w0).permute(1, 2, 3, 0))


@pytest.mark.parametrize('N,H,W,Cin,Cout,k,G', [
    (2, 16, 32, 128, 64, 3, 2),     # halo dgrad, 128 output channels, two statistics groups
    (4, 32, 32, 64, 64, 3, 2),      # halo dgrad, 64 output channels: 16x16 tiles, two rows per tile
    (4, 8, 8, 128, 128, 3, 2),      # halo dgrad on whole 8x8 images (two per tile)
    (2, 14, 14, 128, 64, 3, 2),     # halo dgrad, RAGGED 8x16 tiles, two statistics groups
    (4, 7, 7, 128, 128, 3, 2),      # halo dgrad on whole 7x7 images (two per tile)
    (2, 28, 28, 64, 64, 3, 1),      # halo dgrad, RAGGED 16x16 tiles
    (3, 7, 7, 128, 64, 1, 1),       # generic kernel, ragged M = 147
    (2, 8, 8, 64, 128, 1, 1),       # generic kernel, 64 output channels (32-channel waves)
    (2, 8, 8, 64, 64, 1, 1),        # generic kernel, one K-step
    (3, 7, 7, 64, 512, 1, 1),       # DMA-ring variant with fused statistics: 8 K-steps, 64-channel tile, ragged M = 147
    (4, 8, 8, 256, 320, 1, 2),      # DMA-ring variant with fused statistics: 5 K-steps (odd), four 64-channel tiles, two groups
])
@pytest.mark.parametrize('mask', ['y', 'bits', 'relu', 'none'])
def test_dgrad_fused_bn_backward_statistics(backend, N, H, W, Cin, Cout, k, G, mask):
    """vfs_conv_dgrad_bn: the input gradient is bit-identical to vfs_conv_dgrad, and the partial rows
    {sum g*mask, sum g*mask*xhat} summed per group equal what bn_bwd_reduce computes from the stored
    gradient (fp32 sums: 1e-4 relative)."""
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(N * 7 + Cin + k)
    pad = k // 2
    w = rb(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    _, wd = pack(backend, w)
    dy = d(nhwc(rb(torch.randn(N, Cout, H, W, generator=g))))
    add = d(nhwc(rb(torch.randn(N, Cin, H, W, generator=g))))
    x = rb(torch.randn(N, H, W, Cin, generator=g) * 1.5 + 0.3)           # raw conv output of the producer unit
    M = N * H * W
    mpg = M // G
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    xf = x.float().reshape(G, mpg, Cin)
    mean, var = xf.mean(1), xf.var(1, unbiased=False)
    inv = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma * inv
    bnp = torch.stack([scale, beta - mean * scale, mean, inv], 1).contiguous()    # [G][4][C]
    y = rb(torch.relu(x.float() * scale.repeat_interleave(mpg, 0).reshape(N, H, W, Cin)
                      + (beta - mean * scale).repeat_interleave(mpg, 0).reshape(N, H, W, Cin)
                      + 0.5 * torch.randn(N, H, W, Cin, generator=g)))
    dx0 = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
    lib.conv_dgrad(dy, wd, dx0, add, N, H, W, Cin, H, W, Cout, k, k, 1, pad, None)
    nblk = conv_stats_rows(N, G, H, W, Cout, Cin, k, 1, pad, H, W) * G     # the dgrad as a conv producing [N,H,W,Cin]
    partial = torch.full((nblk, 2, Cin), float('nan'), device=dev)
    dx1 = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
    ymask = d(y.to(torch.bfloat16)) if mask == 'y' else None
    if mask == 'bits':     # the bit-packed form of the same mask (bn_relu = 2)
        ymask = d(pack_relu_mask(y))
    lib.conv_dgrad_bn(dy, wd, dx1, add, d(x.to(torch.bfloat16)), ymask, d(bnp), partial, mpg, {'relu': 1, 'bits': 2}.get(mask, 0),
                      N, H, W, Cin, H, W, Cout, k, k, 1, pad, None)
    assert torch.equal(dx1.cpu(), dx0.cpu())
    if mask == 'bits':     # bit-identical to the 16-byte mask operand
        p2 = torch.full((nblk, 2, Cin), float('nan'), device=dev)
        lib.conv_dgrad_bn(dy, wd, dx1, add, d(x.to(torch.bfloat16)), d(y.to(torch.bfloat16)), d(bnp), p2, mpg, 0,
                          N, H, W, Cin, H, W, Cout, k, k, 1, pad, None)
        assert torch.equal(partial.cpu(), p2.cpu())
    gq = dx0.float().cpu().reshape(G, mpg, Cin).double()
    xq = x.float().reshape(G, mpg, Cin).double()
    if mask in ('y', 'bits'):
        gq = gq * (y.float().reshape(G, mpg, Cin) > 0)
    elif mask == 'relu':
        act = x.float().reshape(G, mpg, Cin) * scale[:, None] + (beta - mean * scale)[:, None]
        gq = gq * (act > 0)
    xh = ((x.float().reshape(G, mpg, Cin) - mean[:, None]) * inv[:, None]).double()
    want1, want2 = gq.sum(1), (gq * xh).sum(1)
    st = partial.cpu().double().reshape(G, nblk // G if G > 1 else nblk, 2, Cin).sum(1)
    assert torch.allclose(st[:, 0], want1, rtol=1e-4, atol=2e-2), (st[:, 0] - want1).abs().max()
    assert torch.allclose(st[:, 1], want2, rtol=1e-4, atol=2e-2), (st[:, 1] - want2).abs().max()


@pytest.mark.parametrize('N,H,W,Cin,Cout,k', [
    (2, 8, 16, 128, 64, 1),      # implicit-GEMM dgrad, one 128-channel tile (64-channel waves)
    (3, 7, 7, 64, 256, 1),       # implicit-GEMM dgrad, 64-channel tile (32-channel waves), DMA ring, ragged M
    (2, 16, 32, 128, 64, 3),     # halo dgrad, 128 output channels
    (2, 32, 32, 64, 64, 3),      # halo dgrad, 64 output channels (16x16 tiles)
    (4, 7, 7, 128, 128, 3),      # halo dgrad on whole 7x7 images
])
def test_dgrad_add_gated_by_bit_packed_mask(backend, N, H, W, Cin, Cout, k):
    """vfs_conv_dgrad_maskadd(add = g, add_mask = bits(y > 0)) == vfs_conv_dgrad(add = g * (y > 0)), bit for bit; the same for
    the statistics-fused entry point"""
    lib, d, dev = backend.lib, backend.d, backend.dev
    gen = torch.Generator().manual_seed(N * 13 + Cin + k)
    pad = k // 2
    w = rb(torch.randn(Cout, Cin, k, k, generator=gen) * (2.0 / (Cin * k * k)) ** 0.5)
    _, wd = pack(backend, w)
    dy = d(nhwc(rb(torch.randn(N, Cout, H, W, generator=gen))))
    g = rb(torch.randn(N, H, W, Cin, generator=gen))
    y = rb(torch.relu(torch.randn(N, H, W, Cin, generator=gen)))
    gm = torch.where(y > 0, g, torch.zeros(()))
    bits = d(pack_relu_mask(y))
    dx0 = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
    dx1 = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
    lib.conv_dgrad(dy, wd, dx0, d(gm.to(torch.bfloat16)), N, H, W, Cin, H, W, Cout, k, k, 1, pad, None)
    lib.conv_dgrad_maskadd(dy, wd, dx1, d(g.to(torch.bfloat16)), bits, N, H, W, Cin, H, W, Cout, k, k, 1, pad, None)
    assert torch.equal(dx0.cpu(), dx1.cpu())
    # with fused BatchNorm-backward statistics
    M = N * H * W
    nblk = conv_stats_rows(N, 1, H, W, Cout, Cin, k, 1, pad, H, W)
    x = rb(torch.randn(N, H, W, Cin, generator=gen))
    bnp = torch.stack([torch.rand(Cin, generator=gen) + 0.5, torch.randn(Cin, generator=gen), torch.randn(Cin, generator=gen) * 0.1,
                       torch.rand(Cin, generator=gen) + 0.5], 0).reshape(1, 4, Cin).contiguous()
    pa, pb = torch.full((nblk, 2, Cin), float('nan'), device=dev), torch.full((nblk, 2, Cin), float('nan'), device=dev)
    dx2 = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
    dx3 = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
    lib.conv_dgrad_bn(dy, wd, dx2, d(gm.to(torch.bfloat16)), d(x.to(torch.bfloat16)), None, d(bnp), pa, M, 1, N, H, W, Cin, H, W, Cout, k, k, 1, pad, None)
    lib.conv_dgrad_bn_maskadd(dy, wd, dx3, d(g.to(torch.bfloat16)), bits, d(x.to(torch.bfloat16)), None, d(bnp), pb, M, 1, N, H, W, Cin, H, W, Cout, k, k,
                              1, pad, None)
    assert torch.equal(dx2.cpu(), dx0.cpu()) and torch.equal(dx3.cpu(), dx0.cpu())
    assert torch.equal(pa.cpu(), pb.cpu())


@pytest.mark.parametrize('N,H,W,Cin,Cout,G', [
    (2, 16, 32, 128, 128, 2),     # 8x16 tiles, two channel chunks, two groups
    (2, 32, 32, 64, 64, 2),       # 16x16 tiles (64 output channels)
    (4, 8, 8, 64, 128, 2),        # whole 8x8 images, two per tile
    (2, 14, 14, 64, 128, 2),      # ragged 8x16 tiles
    (4, 7, 7, 64, 128, 2),        # whole 7x7 images, two per tile
    (2, 28, 28, 64, 64, 1),       # ragged 16x16 tiles (forward) / 8x16 tiles (weight gradient)
])
def test_conv_with_folded_input_batchnorm(backend, N, H, W, Cin, Cout, G):
    """vfs_conv_fwd_bnin / vfs_conv_wgrad_bnin (the activation of the producer unit is never materialised)
    against vfs_bn_act followed by vfs_conv_fwd / vfs_conv_wgrad on the same raw tensor: bit-identical."""
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(N + H + Cin)
    raw = rb(torch.randn(N, H, W, Cin, generator=g) * 1.3).to(torch.bfloat16)
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5)
    wf, _ = pack(backend, w)
    npg = N // G
    bnp = torch.stack([torch.rand(G, Cin, generator=g) + 0.5, torch.randn(G, Cin, generator=g) * 0.4,
                       torch.zeros(G, Cin), torch.ones(G, Cin)], 1).contiguous()
    M = N * H * W
    rawd, bnpd = d(raw), d(bnp)
    act = torch.empty(N, H, W, Cin, dtype=torch.bfloat16, device=dev)
    lib.bn_act(rawd, bnpd, None, None, None, act, M, Cin, M // G, 1, None)
    nblk = conv_stats_rows(N, 1, H, W, Cin, Cout, 3, 1, 1, H, W)
    y0 = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
    y1 = torch.full_like(y0, float('nan'))
    st0 = torch.full((nblk, 2, Cout), float('nan'), device=dev)
    st1 = torch.full_like(st0, float('nan'))
    lib.conv_fwd(act, wf, y0, None, st0, N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, None)
    lib.conv_fwd_bnin(rawd, bnpd, npg, wf, y1, None, st1, N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, None)
    assert torch.equal(y1.cpu(), y0.cpu()) and torch.equal(st1.cpu(), st0.cpu())
    dy = d(rb(torch.randn(N, H, W, Cout, generator=g)).to(torch.bfloat16))
    nsplit, pps = wgrad_splits(M, Cout, 9 * Cin, target_blocks=12, halo_geom=(N, H, W, Cin))
    partial = torch.zeros(nsplit, Cout, 9 * Cin, device=dev)
    g0, g1 = torch.zeros(Cout, Cin, 3, 3, device=dev), torch.zeros(Cout, Cin, 3, 3, device=dev)
    lib.conv_wgrad(dy, act, partial, g0, N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, nsplit, pps, None)
    lib.conv_wgrad_bnin(dy, rawd, bnpd, npg, partial, g1, N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, nsplit, pps, None)
    assert torch.equal(g1.cpu(), g0.cpu())
    tickets = torch.zeros(lib.cfunc('wgrad_tickets')(), dtype=torch.int32, device=dev)
    g2 = torch.zeros(Cout, Cin, 3, 3, device=dev)
    lib.conv_wgrad_inl(dy, rawd, bnpd, npg, partial, g2, tickets, N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, nsplit, pps, None)      # in-launch reduction
    assert relerr(g2.cpu(), g0.cpu()) < 2e-6 and int(tickets.cpu().abs().sum()) == 0
    with pytest.raises(Exception):       # only the halo-tile shapes fold the input BatchNorm
        lib.conv_fwd_bnin(rawd, bnpd, npg, wf, y1, None, st1, N, H, W, Cin, H // 2, W // 2, Cout, 3, 3, 2, 1, None)


@pytest.mark.parametrize('N,H,W,Cin,Cout,G', [
    (2, 8, 16, 64, 256, 2),       # one K-step (the single-buffer pipeline), 128-channel tiles, two groups
    (4, 8, 8, 128, 512, 2),       # two K-steps, two groups of 128 pixels
    (2, 16, 16, 256, 128, 1),     # four K-steps on the double-buffer pipeline
    (2, 8, 8, 64, 64, 1),         # 64-channel tile
])
def test_conv1x1_with_folded_input_batchnorm(backend, N, H, W, Cin, Cout, G):
    """round 6: vfs_conv_fwd_bnin / vfs_conv_wgrad_bnin for the 1x1 / stride-1 consumer (the conv2 -> conv3 edge of a bottleneck
    block) against vfs_bn_act followed by vfs_conv_fwd / vfs_conv_wgrad on the same raw tensor: bit-identical."""
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(N + H + Cin + 1)
    raw = rb(torch.randn(N, H, W, Cin, generator=g) * 1.3).to(torch.bfloat16)
    w = rb(torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5)
    wf, _ = pack(backend, w)
    npg = N // G
    bnp = torch.stack([torch.rand(G, Cin, generator=g) + 0.5, torch.randn(G, Cin, generator=g) * 0.4,
                       torch.zeros(G, Cin), torch.ones(G, Cin)], 1).contiguous()
    M = N * H * W
    rawd, bnpd = d(raw), d(bnp)
    act = torch.empty(N, H, W, Cin, dtype=torch.bfloat16, device=dev)
    lib.bn_act(rawd, bnpd, None, None, None, act, M, Cin, M // G, 1, None)
    nblk = conv_stats_rows(N, 1, H, W, Cin, Cout, 1, 1, 0, H, W)
    y0 = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
    y1 = torch.full_like(y0, float('nan'))
    st0 = torch.full((nblk, 2, Cout), float('nan'), device=dev)
    st1 = torch.full_like(st0, float('nan'))
    lib.conv_fwd(act, wf, y0, None, st0, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, None)
    lib.conv_fwd_bnin(rawd, bnpd, npg, wf, y1, None, st1, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, None)
    assert torch.equal(y1.cpu(), y0.cpu()) and torch.equal(st1.cpu(), st0.cpu())
    dy = d(rb(torch.randn(N, H, W, Cout, generator=g)).to(torch.bfloat16))
    nsplit, pps = wgrad_splits(M, Cout, Cin, target_blocks=12)
    partial = torch.zeros(nsplit, Cout, Cin, device=dev)
    g0, g1 = torch.zeros(Cout, Cin, 1, 1, device=dev), torch.zeros(Cout, Cin, 1, 1, device=dev)
    lib.conv_wgrad(dy, act, partial, g0, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, nsplit, pps, None)
    lib.conv_wgrad_bnin(dy, rawd, bnpd, npg, partial, g1, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, nsplit, pps, None)
    assert torch.equal(g1.cpu(), g0.cpu())


BIG_CASES = [  # the layer shapes of the bench configs (per-GPU batch reduced), ragged M included
    (8, 64, 64, 64, 64, 3, 1, 1),      # R18 layer1 @256
    (8, 64, 64, 64, 128, 3, 2, 1),     # R18 layer2.0.conv1
    (8, 16, 16, 256, 512, 3, 2, 1),    # layer4.0.conv1
    (16, 8, 8, 512, 512, 3, 1, 1),     # layer4
    (4, 64, 64, 64, 256, 1, 1, 0),     # R50 layer1 conv3
    (4, 32, 32, 512, 1024, 1, 2, 0),   # R50 layer3 downsample
    (3, 60, 107, 64, 64, 3, 1, 1),     # DAVIS-sized feature map, ragged pixel tiles
]


@pytest.mark.gpu
@pytest.mark.parametrize('N,H,W,Cin,Cout,k,stride,pad', BIG_CASES)
def test_conv_bench_shapes(gpu_backend, N, H, W, Cin, Cout, k, stride, pad):
    run_conv_case(gpu_backend, N, H, W, Cin, Cout, k, stride, pad, wgrad_blocks=1024)


def test_stem_fwd_wgrad(backend):
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(5)
    N, H, W = (2, 20, 18) if backend.name == 'emu' else (6, 96, 128)
    x = rb(torch.randn(N, 3, H, W, generator=g))
    w = rb(torch.randn(64, 3, 7, 7, generator=g) * 0.1)
    wf, _ = pack(backend, w, stem=True)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    # reference input layout [B][V][3][T][H][W] with B=N, V=1, T=1
    x4 = torch.full((N, H, W, 4), float('nan'), dtype=torch.bfloat16, device=dev)
    lib.imgs_to_nhwc4(d(x.reshape(N, 1, 3, 1, H, W).contiguous()), x4, N, 1, 1, H, W, W, None)
    assert torch.equal(x4[..., :3].float().cpu(), x.permute(0, 2, 3, 1)) and (x4[..., 3] == 0).all()
    M = N * Ho * Wo
    y = torch.full((N, Ho, Wo, 64), float('nan'), dtype=torch.bfloat16, device=dev)
    ntile = N * ((Ho + 7) // 8) * ((Wo + 15) // 16)
    stats = torch.full((ntile, 2, 64), float('nan'), device=dev)
    ref = F.conv2d(x, w, None, 2, 3)
    # default grid (one tile per workgroup at this size), then ONE workgroup walking every tile and
    # three walking ragged ranges: the statistics run must break at every image boundary
    for blocks in (0, 1, 3):
        y.fill_(float('nan'))
        stats.fill_(float('nan'))
        lib.set_option(b'stem_blocks', blocks)
        try:
            lib.stem_fwd(x4, wf, y, stats, N, H, W, Ho, Wo, None)
        finally:
            lib.set_option(b'stem_blocks', 0)
        assert relerr(nchw(y.cpu()), ref) < 6e-3
        yf = y.float().cpu().reshape(M, 64).double()
        st = stats.cpu().double()
        assert torch.allclose(st[:, 0].sum(0), yf.sum(0), rtol=1e-4, atol=5e-3)
        assert torch.allclose(st[:, 1].sum(0), (yf * yf).sum(0), rtol=1e-4, atol=5e-3)
        half = ntile // 2     # first / second half of the images = first / second half of the rows
        assert torch.allclose(st[:half, 0].sum(0), yf[:M // 2].sum(0), rtol=1e-4, atol=5e-3)
    # the generic implicit-GEMM stem path must agree bit-for-bit on the output
    lib.set_option(b'stem_direct', 0)
    y2 = torch.full((N, Ho, Wo, 64), float('nan'), dtype=torch.bfloat16, device=dev)
    lib.stem_fwd(x4, wf, y2, None, N, H, W, Ho, Wo, None)
    lib.set_option(b'stem_direct', 1)
    assert relerr(nchw(y2.cpu()), ref) < 6e-3
    dy = rb(torch.randn(N, 64, Ho, Wo, generator=g))
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, 2, 3).backward(dy)
    nsplit, pps = wgrad_splits(M, 64, 256, target_blocks=6 if backend.name == 'emu' else 256)
    partial = torch.zeros(nsplit, 64, 256, device=dev)
    grad = torch.zeros(64, 3, 7, 7, device=dev)
    lib.stem_wgrad(d(nhwc(dy)), x4, partial, grad, N, H, W, Ho, Wo, nsplit, pps, None)
    assert relerr(grad.cpu(), wr.grad) < 3e-4


@pytest.mark.parametrize('M,Cin,Cout', [(64, 512, 256), (200, 256, 192), (8, 1024, 128)])
def test_splitk_linear_matches_plain_kernel(backend, M, Cin, Cout):
    """the head's Linear layers (1x1 conv on 1x1 'images'): split-K forward / dgrad = the plain kernel up to the
    fp32 summation order, statistics rows included; the tickets come back to zero (second launch works)"""
    from vfs_amd.packing import igemm_ksplit
    lib, d = backend.lib, backend.d
    g = torch.Generator().manual_seed(M + Cin)
    x = rb(torch.randn(M, Cin, 1, 1, generator=g))
    w = rb(torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5)
    bias = torch.randn(Cout, generator=g)
    wf, wd = pack(backend, w)
    xh = d(nhwc(x))
    ks, need = igemm_ksplit(M, Cout, Cin)
    assert ks > 1 and need > 1024
    ws = torch.zeros(need, device=backend.dev)
    nblk = (M + 127) // 128
    outs = []
    for which in ('plain', 'split', 'split'):
        y = torch.full((M, 1, 1, Cout), float('nan'), dtype=torch.bfloat16, device=backend.dev)
        st = torch.full((nblk, 2, Cout), float('nan'), device=backend.dev)
        if which == 'plain':
            lib.conv_fwd(xh, wf, y, d(bias), st, M, 1, 1, Cin, 1, 1, Cout, 1, 1, 1, 0, None)
        else:
            lib.conv_fwd_splitk(xh, wf, y, d(bias), st, ws, ks, M, 1, 1, Cin, 1, 1, Cout, 1, 1, 1, 0, None)
        outs.append((y.float().cpu(), st.cpu()))
    assert torch.equal(ws[:1024].cpu(), torch.zeros(1024))
    ref = F.conv2d(x, w, bias)
    for y, st in outs:
        assert relerr(nchw(y), ref) < 6e-3
    assert relerr(outs[1][0], outs[0][0]) < 4e-3 and torch.equal(outs[1][0], outs[2][0])     # deterministic
    assert torch.allclose(outs[1][1], outs[0][1], rtol=2e-2, atol=0.5) and torch.equal(outs[1][1], outs[2][1])
    # dgrad with the residual-gradient add
    dy = rb(torch.randn(M, Cout, 1, 1, generator=g))
    add = rb(torch.randn(M, Cin, 1, 1, generator=g))
    ks2, need2 = igemm_ksplit(M, Cin, Cout)
    if ks2 > 1:
        ws2 = torch.zeros(need2, device=backend.dev)
        dx = torch.full((M, 1, 1, Cin), float('nan'), dtype=torch.bfloat16, device=backend.dev)
        lib.conv_dgrad_splitk(d(nhwc(dy)), wd, dx, d(nhwc(add)), ws2, ks2, M, 1, 1, Cin, 1, 1, Cout, 1, 1, 1, 0, None)
        want = torch.einsum('mo,oi->mi', dy[:, :, 0, 0], w[:, :, 0, 0]) + add[:, :, 0, 0]
        assert relerr(dx.float().cpu()[:, 0, 0], want) < 6e-3
        assert torch.equal(ws2[:1024].cpu(), torch.zeros(1024))


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride,dil', [(2, 12, 16, 64, 128, 1, 2), (1, 16, 16, 128, 64, 1, 4), (2, 13, 13, 64, 64, 2, 2)])
def test_dilated_forward_conv(backend, N, H, W, Cin, Cout, stride, dil):
    """vfs_conv_fwd_dilated (3x3, padding = dilation: the frozen dilated backbone of the SiamFC probe) vs torch conv2d"""
    lib, d = backend.lib, backend.d
    g = torch.Generator().manual_seed(N + H + dil)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5)
    wf, _ = pack(backend, w)
    Ho, Wo = (H + 2 * dil - 2 * dil - 1) // stride + 1, (W + 2 * dil - 2 * dil - 1) // stride + 1
    y = torch.full((N, Ho, Wo, Cout), float('nan'), dtype=torch.bfloat16, device=backend.dev)
    lib.conv_fwd_dilated(d(nhwc(x)), wf, y, None, None, N, H, W, Cin, Ho, Wo, Cout, 3, 3, stride, dil, dil, None)
    ref = F.conv2d(x, w, None, stride, dil, dil)
    assert relerr(nchw(y.cpu()), ref) < 6e-3
    with pytest.raises(Exception):      # the output size must follow the dilated formula
        lib.conv_fwd_dilated(d(nhwc(x)), wf, y, None, None, N, H, W, Cin, Ho + 1, Wo, Cout, 3, 3, stride, dil, dil, None)


def _sweep_shapes(n, seed):
    """pseudo-random conv problems around the dispatch boundaries (ragged tiles, 7/8/14-pixel maps, odd batch sizes,
    64/128/192 channels, 1x1 and 3x3, stride 1 and 2); fixed seed, so the set is reproducible"""
    import random
    rnd = random.Random(seed)
    out = []
    while len(out) < n:
        k = rnd.choice([1, 3, 3])
        stride = rnd.choice([1, 1, 2])
        shape = (rnd.choice([1, 2, 3, 4]), rnd.choice([5, 7, 8, 9, 12, 14, 16, 20]), rnd.choice([6, 7, 8, 14, 16, 17, 24, 28]),
                 rnd.choice([64, 128]), rnd.choice([64, 128, 192]), k, stride, 1 if k == 3 else 0)
        if shape not in out:
            out.append(shape)
    return out


@pytest.mark.parametrize('N,H,W,Cin,Cout,k,stride,pad', _sweep_shapes(28, 20260927))
def test_conv_shape_sweep(backend, N, H, W, Cin, Cout, k, stride, pad):
    """forward (+ statistics rows), dgrad (+ residual add) and wgrad of every sampled problem vs torch: exercises the
    host mirrors of the tiling rules (conv_halo_eligible / conv_stats_rows / wgrad_splits) together with the kernels"""
    run_conv_case(backend, N, H, W, Cin, Cout, k, stride, pad)


def test_ring_upfront_reads_bit_identical(backend):
    """option igemm_ring_upfront (all fragment reads of a K-step before its MFMAs; prepared for the next round, off by
    default): same MFMA order, so the outputs must be bit-identical to the default schedule"""
    lib, d = backend.lib, backend.d
    g = torch.Generator().manual_seed(5)
    N, H, W, Cin, Cout = 2, 8, 8, 512, 256
    x = d(nhwc(rb(torch.randn(N, Cin, H, W, generator=g))))
    w = rb(torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5)
    wf, wd = pack(backend, w)
    outs = []
    for flag in (0, 1):
        lib.set_option(b'igemm_ring_upfront', flag)
        lib.set_option(b'igemm_ring_mfma32', 0)      # (the default ring runs 32x32x16 MFMAs: another K grouping on the hardware)
        try:
            y = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=backend.dev)
            st = torch.full((1, 2, Cout), float('nan'), device=backend.dev)
            lib.conv_fwd(x, wf, y, None, st, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, None)
            outs.append((y.cpu(), st.cpu()))
        finally:
            lib.set_option(b'igemm_ring_upfront', 0)
            lib.set_option(b'igemm_ring_mfma32', 1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert relerr(nchw(outs[1][0]), F.conv2d(nchw(x.cpu()), w)) < 6e-3


@pytest.mark.parametrize('N,H,W,Cin,Cout', [
    (4, 8, 8, 256, 128),      # whole 8x8 images in pairs, four channel chunks: the ring wraps across chunk boundaries, two patch buffers
    (2, 16, 16, 128, 256),    # 8x16 tiles, two chunks, two channel blocks
    (2, 14, 14, 192, 128),    # ragged tiles, three chunks
    (3, 7, 7, 64, 128),       # one chunk only (nine steps), odd image count (the second image of the last tile does not exist)
])
def test_halo_deep_schedule_matches_two_stage(backend, N, H, W, Cin, Cout):
    """the deep schedule of the halo kernel (32x32x16 MFMAs, four weight stages requested four taps ahead, the patch by LDS-DMA
    into two buffers, next tap's fragments read under this tap's MFMAs; option halo_deep_max) against the two-stage
    schedule: the same products, K summed 16 instead of 32 at a time - equal to fp32 rounding (the emulator's MFMAs are
    k-ordered fmaf chains: bit-identical there), statistics rows consistent with the stored outputs, both against torch"""
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(N * 3 + Cin + Cout)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5)
    wf, wd = pack(backend, w)
    dy = rb(torch.randn(N, Cout, H, W, generator=g))
    add = rb(torch.randn(N, Cin, H, W, generator=g))
    nblk = conv_stats_rows(N, 1, H, W, Cin, Cout, 3, 1, 1, H, W)
    outs = []
    for deep in (256, 0):
        lib.set_option(b'halo_deep_max', deep)
        try:
            y = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
            stats = torch.full((nblk, 2, Cout), float('nan'), device=dev)
            lib.conv_fwd(d(nhwc(x)), wf, y, None, stats, N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, None)
            outs.append((y.cpu(), stats.cpu()))
            if Cin % 128 == 0:      # the dgrad produces Cin channels: a 128-channel tile again
                dx = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
                lib.conv_dgrad(d(nhwc(dy)), wd, dx, d(nhwc(add)), N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, None)
                outs[-1] += (dx.cpu(),)
        finally:
            lib.set_option(b'halo_deep_max', 256)
    for a_, b_ in zip(outs[0], outs[1]):
        if backend.name == 'emu':
            assert torch.equal(a_, b_)
        else:      # one bf16 ulp where an fp32 rounding difference crosses a rounding boundary; fp32 statistics rows of those outputs
            assert relerr(a_.float(), b_.float()) < 8e-3
    yf = outs[0][0].float().reshape(-1, Cout).double()
    assert torch.allclose(outs[0][1][:, 0].double().sum(0), yf.sum(0), rtol=1e-4, atol=5e-3)
    assert torch.allclose(outs[0][1][:, 1].double().sum(0), (yf * yf).sum(0), rtol=1e-4, atol=5e-3)
    assert relerr(nchw(outs[0][0]), F.conv2d(x, w, None, 1, 1)) < 6e-3
    if Cin % 128 == 0:
        xr = x.clone().requires_grad_(True)
        F.conv2d(xr, w, None, 1, 1).backward(dy)
        assert relerr(nchw(outs[0][2]), xr.grad + add) < 6e-3


@pytest.mark.parametrize('N,H,W,Cin,Cout,k,L', [
    (4, 16, 32, 64, 128, 1, 2),       # implicit-GEMM kernel: 16 rows, groups of 4, one 128-channel tile
    (3, 16, 16, 64, 192, 1, 1),       # 6 rows in groups of 2, three 64-channel tiles
    (5, 16, 16, 128, 64, 1, 3),       # 10 rows, groups of 8: the last group holds two rows
    (4, 16, 32, 64, 128, 3, 2),       # halo kernel, 8x16 tiles (one row per workgroup): 16 rows, groups of 4
    (2, 32, 32, 64, 64, 3, 2),        # halo kernel, 16x16 tiles (TWO rows per workgroup): 16 rows, groups of 4 = 2 workgroups
    (6, 8, 8, 64, 128, 3, 1),         # halo kernel, whole 8x8 images in pairs
])
def test_conv_forward_coarse_statistics_rows(backend, N, H, W, Cin, Cout, k, L):
    """vfs_conv_fwd_coarse: same outputs and fine rows as vfs_conv_fwd, and coarse row g = the fine rows 2^L g .. added in row
    order (fp32, deterministic) by the last workgroup of the group to arrive; the tickets are back at zero (run twice)"""
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(N + H + Cin + Cout + k)
    pad = k // 2
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    wf, _ = pack(backend, w)
    nblk = conv_stats_rows(N, 1, H, W, Cin, Cout, k, 1, pad, H, W)
    ng = (nblk + (1 << L) - 1) >> L
    y0 = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
    s0 = torch.full((nblk, 2, Cout), float('nan'), device=dev)
    lib.conv_fwd(d(nhwc(x)), wf, y0, None, s0, N, H, W, Cin, H, W, Cout, k, k, 1, pad, None)
    tickets = torch.zeros(ng * ((Cout + 63) // 64), dtype=torch.int32, device=dev)
    for rep in range(2):
        y1 = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
        s1 = torch.full((nblk, 2, Cout), float('nan'), device=dev)
        c1 = torch.full((ng, 2, Cout), float('nan'), device=dev)
        lib.conv_fwd_coarse(d(nhwc(x)), wf, y1, None, s1, c1, tickets, L, N, H, W, Cin, H, W, Cout, k, k, 1, pad, None)
        assert torch.equal(y1.cpu(), y0.cpu()) and torch.equal(s1.cpu(), s0.cpu())
        assert int(tickets.cpu().abs().sum()) == 0
        want = torch.zeros(ng, 2, Cout)
        fine = s0.cpu()
        for r in range(nblk):      # row order, fp32
            want[r >> L] = want[r >> L] + fine[r]
        assert torch.equal(c1.cpu(), want), float((c1.cpu() - want).abs().max())


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 8, 8, 512, 256), (3, 7, 7, 1024, 64), (1, 5, 5, 320, 128), (4, 8, 8, 256, 512)])
def test_ring_on_32x32_mfma(backend, N, H, W, Cin, Cout):
    """the pure-GEMM DMA ring on v_mfma_f32_32x32x16_bf16 (PIPE 5: lean DMA issue, accumulators handed to the shared epilogue
    through an LDS transposition) against the 16x16x32 ring: forward (+ statistics rows) and dgrad (+ residual) equal to fp32
    rounding of another K grouping (bit-identical on the emulator, whose MFMAs are k-ordered fmaf chains), both against torch"""
    lib, d, dev = backend.lib, backend.d, backend.dev
    g = torch.Generator().manual_seed(N + Cin + Cout)
    x = rb(torch.randn(N, Cin, H, W, generator=g))
    w = rb(torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5)
    wf, wd = pack(backend, w)
    dy = rb(torch.randn(N, Cout, H, W, generator=g))
    add = rb(torch.randn(N, Cin, H, W, generator=g))
    M = N * H * W
    outs = []
    for flag in (1, 0):
        lib.set_option(b'igemm_ring_mfma32', flag)
        try:
            y = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
            st = torch.full(((M + 127) // 128, 2, Cout), float('nan'), device=dev)
            lib.conv_fwd(d(nhwc(x)), wf, y, None, st, N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, None)
            res = [y.float().cpu(), st.cpu()]
            if Cout % 64 == 0 and Cout >= 256:
                dx = torch.full((N, H, W, Cin), float('nan'), dtype=torch.bfloat16, device=dev)
                lib.conv_dgrad(d(nhwc(dy)), wd, dx, d(nhwc(add)), N, H, W, Cin, H, W, Cout, 1, 1, 1, 0, None)
                res.append(dx.float().cpu())
            outs.append(res)
        finally:
            lib.set_option(b'igemm_ring_mfma32', 1)
    for a_, b_ in zip(outs[0], outs[1]):
        if backend.name == 'emu':
            assert torch.equal(a_, b_)
        else:
            assert torch.isfinite(a_).all() and relerr(a_, b_) < 8e-3
    assert relerr(nchw(outs[0][0].to(torch.bfloat16)), F.conv2d(x, w)) < 6e-3
    yf = outs[0][0].reshape(M, Cout).double()
    assert torch.allclose(outs[0][1][:, 0].double().sum(0), yf.sum(0), rtol=1e-4, atol=5e-3)
    if len(outs[0]) > 2:
        xr = x.clone().requires_grad_(True)
        F.conv2d(xr, w).backward(dy)
        assert relerr(nchw(outs[0][2].to(torch.bfloat16)), xr.grad + add) < 6e-3
