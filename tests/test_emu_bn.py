"""BatchNorm / pooling / loss / SGD kernels vs torch (CPU fp32) on the same bf16-rounded operands.
backend=emu: fiber emulator on the CPU; backend=gpu: libvfs_hip.so on the MI355X."""
import pytest
import torch
import torch.nn.functional as F

from oracle import vfs_oracle as O
from tests.emu_util import nchw, nhwc, pack_relu_mask, rb, relerr


def bn_forward_chain(lib, x_nhwc, gamma, beta, G, rm, rv):
    """stats from a [M][C] bf16 tensor via the same kernels the engine chains after a conv."""
    M, C = x_nhwc.reshape(-1, x_nhwc.shape[-1]).shape
    mpg = M // G
    ppb = 128 if mpg % 128 == 0 else mpg
    # emulate the conv epilogue partials with torch (the conv test checks the real ones)
    xf = x_nhwc.float().reshape(M, C)
    nblk = M // ppb
    partial = torch.stack([torch.stack([xf[b * ppb:(b + 1) * ppb].sum(0), (xf[b * ppb:(b + 1) * ppb] ** 2).sum(0)])
                           for b in range(nblk)])
    sums = torch.zeros(G, 2, C, dtype=torch.float64)
    lib.bn_reduce_partials(partial, sums, torch.zeros(G * 128 * 2 * C, dtype=torch.float64), G, nblk // G, C, None)
    bnp = torch.zeros(G, 4, C)
    lib.bn_finalize(sums, gamma, beta, bnp, rm, rv, G, C, float(mpg), 1e-5, 0.1, None)
    return bnp, mpg


def test_bn_train_forward_two_groups_and_residuals(backend):
    lib = backend.hostlib
    g = torch.Generator().manual_seed(0)
    N, C, H, W, G = 4, 64, 8, 8, 2
    x = rb(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    rm, rv = torch.zeros(C), torch.ones(C)
    bnp, mpg = bn_forward_chain(lib, nhwc(x), gamma, beta, G, rm, rv)
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data.copy_(gamma); bn.bias.data.copy_(beta)
    bn.train()
    refs = [bn(x[:2]), bn(x[2:])]            # two separate BN calls, like the two views
    ref = torch.cat(refs).detach()
    assert relerr(rm, bn.running_mean) < 1e-5 and relerr(rv, bn.running_var) < 1e-5
    M = N * H * W
    y = torch.empty(N, H, W, C, dtype=torch.bfloat16)
    lib.bn_act(nhwc(x), bnp, None, None, None, y, M, C, mpg, 1, None)
    assert relerr(nchw(y), F.relu(ref)) < 6e-3
    res = rb(torch.randn(N, C, H, W, generator=g))
    lib.bn_act(nhwc(x), bnp, nhwc(res), None, None, y, M, C, mpg, 1, None)
    assert relerr(nchw(y), F.relu(ref + res)) < 6e-3
    lib.bn_act(nhwc(x), bnp, None, nhwc(res), bnp, y, M, C, mpg, 0, None)   # raw residual with its own BN
    ref_r = torch.cat([bn(res[:2]), bn(res[2:])]).detach()   # NOTE: uses res's own batch stats
    bnp_r, _ = bn_forward_chain(lib, nhwc(res), gamma, beta, G, torch.zeros(C), torch.ones(C))
    lib.bn_act(nhwc(x), bnp, None, nhwc(res), bnp_r, y, M, C, mpg, 0, None)
    assert relerr(nchw(y), ref + ref_r) < 6e-3
    # eval-mode parameters
    bnp_e = torch.zeros(1, 4, C)
    lib.bn_eval_params(gamma, beta, bn.running_mean, bn.running_var, bnp_e, C, 1e-5, None)
    bn.eval()
    lib.bn_act(nhwc(x), bnp_e, None, None, None, y, M, C, M, 0, None)
    assert relerr(nchw(y), bn(x).detach()) < 6e-3


def test_bn_backward_matches_autograd(backend):
    lib = backend.hostlib
    g = torch.Generator().manual_seed(1)
    N, C, H, W, G = 4, 128, 4, 8, 2
    x = rb(torch.randn(N, C, H, W, generator=g) * 2 + 0.5)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    bnp, mpg = bn_forward_chain(lib, nhwc(x), gamma, beta, G, torch.zeros(C), torch.ones(C))
    M = N * H * W
    y = torch.empty(N, H, W, C, dtype=torch.bfloat16)
    lib.bn_act(nhwc(x), bnp, None, None, None, y, M, C, mpg, 1, None)
    gout = rb(torch.randn(N, C, H, W, generator=g))
    # autograd reference per group, on the same rounded tensors
    xs = x.clone().requires_grad_(True)
    gm_, bt_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    outs = [F.relu(F.batch_norm(xs[i:i + 2], None, None, gm_, bt_, True, 0.1, 1e-5)) for i in (0, 2)]
    torch.cat(outs).backward(gout * (nchw(y) > 0))   # mask by the STORED activation like the kernel
    ppb = 32
    nblk = M // ppb
    partial = torch.zeros(nblk, 2, C)
    scratch = torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64)   # ticket counters + chunk sums
    for ymask, relu in ((y, 0), (None, 1)):       # mask from the stored output / recomputed from x
        lib.bn_bwd_reduce(nhwc(gout), ymask, nhwc(x), bnp, partial, M, C, mpg, ppb, relu, None)
        sums = torch.zeros(G, 2, C, dtype=torch.float64)
        lib.bn_reduce_partials(partial, sums, scratch, G, nblk // G, C, None)
        dx = torch.empty(N, H, W, C, dtype=torch.bfloat16)
        gmask = torch.empty(N, H, W, C, dtype=torch.bfloat16)
        lib.bn_bwd_apply(nhwc(gout), ymask, nhwc(x), bnp, sums, dx, gmask, M, C, mpg, float(mpg), relu, None)
        assert relerr(nchw(dx), xs.grad) < 8e-3
        assert torch.equal(nchw(gmask), gout * (nchw(y) > 0))
    dgamma, dbeta = torch.zeros(C), torch.zeros(C)
    lib.bn_param_grad(sums, dgamma, dbeta, G, C, None)
    assert relerr(dgamma, gm_.grad) < 1e-4 and relerr(dbeta, bt_.grad) < 1e-4


@pytest.mark.parametrize('G,bpg,C', [(2, 300, 96), (1, 5000, 64), (2, 65, 2048)])
def test_bn_chunked_single_launch_reduction(backend, G, bpg, C):
    """large row counts: many workgroups reduce row chunks and the one that draws the last ticket of a
    channel block finishes it - sums, fused finalize and fused parameter gradients against fp64 torch,
    twice in a row on the same scratch (the tickets must return to zero)"""
    lib = backend.hostlib
    g = torch.Generator().manual_seed(G * 1000 + bpg)
    partial = torch.randn(G * bpg, 2, C, generator=g)
    partial[:, 1].abs_().add_(partial[:, 0] ** 2)       # sum of squares >= (sum)^2 / n
    ref = partial.double().reshape(G, bpg, 2, C).sum(1)
    scratch = torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    count = float(bpg * 128)
    for _ in range(2):
        sums = torch.zeros(G, 2, C, dtype=torch.float64)
        lib.bn_reduce_partials(partial, sums, scratch, G, bpg, C, None)
        assert torch.allclose(sums, ref, rtol=1e-12, atol=1e-9)
        assert (scratch[:32].view(torch.int32) == 0).all()
        sums2 = torch.zeros(G, 2, C, dtype=torch.float64)
        bnp, rm, rv = torch.zeros(G, 4, C), torch.zeros(C), torch.ones(C)
        lib.bn_stats_finalize(partial, sums2, scratch, gamma, beta, bnp, rm, rv, G, bpg, C, count, 1e-5, 0.1, None)
        assert torch.equal(sums2, sums)
        bnp1, rm1, rv1 = torch.zeros(G, 4, C), torch.zeros(C), torch.ones(C)
        lib.bn_finalize(sums, gamma, beta, bnp1, rm1, rv1, G, C, count, 1e-5, 0.1, None)
        assert torch.equal(bnp, bnp1) and torch.equal(rm, rm1) and torch.equal(rv, rv1)
        dg, db = torch.ones(C), torch.ones(C)
        sums3 = torch.zeros(G, 2, C, dtype=torch.float64)
        lib.bn_bwd_sums_paramgrad(partial, sums3, scratch, dg, db, G, bpg, C, None)
        assert torch.equal(sums3, sums)
        assert torch.allclose(db.double() - 1, ref[:, 0].sum(0), rtol=1e-5, atol=1e-3)
        assert torch.allclose(dg.double() - 1, ref[:, 1].sum(0), rtol=1e-5, atol=1e-3)


def test_stem_bn_relu_maxpool_fwd_bwd(backend):
    lib = backend.hostlib
    g = torch.Generator().manual_seed(2)
    N, C, H, W = 2, 64, 10, 12
    x = rb(torch.randn(N, C, H, W, generator=g))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    bnp, _ = bn_forward_chain(lib, nhwc(x), gamma, beta, 1, torch.zeros(C), torch.ones(C))
    Hp, Wp = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(N, Hp, Wp, C, dtype=torch.bfloat16)
    idx = torch.empty(N, Hp, Wp, C, dtype=torch.uint8)
    lib.bn_relu_maxpool(nhwc(x), bnp, y, idx, None, N, H, W, C, Hp, Wp, N, None)
    a = rb(F.relu(F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5)))
    a.requires_grad_(True)
    ref = F.max_pool2d(a, 3, 2, 1)
    # the kernel pools bf16(relu(bn(x))) computed with its own fp32 scale/shift: allow 1 bf16 ulp
    assert relerr(nchw(y), ref) < 8e-3
    gp = rb(torch.randn(N, C, Hp, Wp, generator=g))
    ref.backward(gp)
    ga = torch.empty(N, H, W, C, dtype=torch.bfloat16)
    lib.maxpool_relu_bwd(nhwc(gp), y, idx, ga, N, H, W, C, Hp, Wp, None)
    want = a.grad * (a.detach() > 0)
    got = nchw(ga)
    # argmax ties between equal bf16 activations may route differently only if the kernel's
    # activation differs by an ulp from torch's; require near-total agreement
    frac = ((got - want).abs() > 1e-2 * want.abs().max()).float().mean()
    assert frac < 2e-3, frac


def test_avgpool_bias_loss_sgd(backend):
    lib = backend.hostlib
    g = torch.Generator().manual_seed(3)
    N, HW, C = 8, 6, 128
    x = rb(torch.randn(N, HW, C, generator=g))
    y = torch.empty(N, C, dtype=torch.bfloat16)
    lib.avgpool_fwd(x.to(torch.bfloat16), y, N, HW, C, None)
    assert relerr(y.float(), x.mean(1)) < 6e-3
    gy = rb(torch.randn(N, C, generator=g))
    gx = torch.empty(N, HW, C, dtype=torch.bfloat16)
    lib.avgpool_bwd(gy.to(torch.bfloat16), gx, N, HW, C, None)
    assert relerr(gx.float(), (gy / HW)[:, None, :].expand(N, HW, C)) < 6e-3
    db = torch.ones(C)
    lib.bias_grad(gy.to(torch.bfloat16), db, N, C, None)
    assert relerr(db - 1, gy.sum(0)) < 1e-5

    # cosine loss with temporal rolls (T=4, K=4) and without (T=1)
    for T, K in ((4, 4), (1, 1), (4, 1)):
        p1, z1, p2, z2 = (rb(torch.randn(N, C, generator=g)) for _ in range(4))
        w = 1.0 / T if K > 1 else 1.0
        loss = torch.empty(K, N)
        args = [t.to(torch.bfloat16) for t in (p1, z1, p2, z2)]
        lib.cosine_loss_fwd(*args, loss, N, C, T, K, 0, w, None)
        p1r, p2r = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
        refs = [O.head_loss(p1r, z1, p2r, z2, w)]
        if K > 1:
            z2v, p2v = O.images2video(z2, T), O.images2video(p2r, T)
            for i in range(1, T):
                refs.append(O.head_loss(p1r, z1, O.video2images(p2v.roll(i, dims=2)),
                                        O.video2images(z2v.roll(i, dims=2)), w))
        ref = torch.stack(refs)
        assert relerr(loss, ref.detach()) < 1e-5, (T, K)
        gl = torch.randn(K, N, generator=g)
        (ref * gl).sum().backward()
        dp1 = torch.empty(N, C, dtype=torch.bfloat16)
        dp2 = torch.empty(N, C, dtype=torch.bfloat16)
        lib.cosine_loss_bwd(*args, gl, dp1, dp2, N, C, T, K, 0, w, None)
        assert relerr(dp1.float(), p1r.grad) < 6e-3 and relerr(dp2.float(), p2r.grad) < 6e-3, (T, K)

    # SGD, three steps against torch.optim.SGD
    n = 1027
    p = torch.randn(n, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.05, momentum=0.9, weight_decay=1e-4)
    buf = torch.zeros(n)
    for _ in range(3):
        gr = torch.randn(n, generator=g)
        pr.grad = gr.clone()
        opt.step()
        lib.sgd_step(p, gr, buf, n, 0.05, 0.9, 1e-4, None, None)
    assert relerr(p, pr.detach()) < 1e-6
    # the device-side guard: a non-zero skip word (the SyncBN exchange's error flag) leaves weights and momentum untouched
    before, bbefore = p.clone(), buf.clone()
    lib.sgd_step(p, gr, buf, n, 0.05, 0.9, 1e-4, torch.ones(1, dtype=torch.int64, device=p.device), None)
    assert torch.equal(p, before) and torch.equal(buf, bbefore)
    lib.sgd_step(p, gr, buf, n, 0.05, 0.9, 1e-4, torch.zeros(1, dtype=torch.int64, device=p.device), None)
    assert not torch.equal(p, before)


def test_bn_reduce_partials_two_stage(backend):
    """row counts above 64 take the chunked two-stage path; result must equal the fp64 column sums"""
    lib = backend.hostlib
    g = torch.Generator().manual_seed(4)
    G, bpg, C = 2, 1000, 96
    partial = torch.randn(G * bpg, 2, C, generator=g)
    sums = torch.zeros(G, 2, C, dtype=torch.float64)
    lib.bn_reduce_partials(partial, sums, torch.zeros(G * 128 * 2 * C, dtype=torch.float64), G, bpg, C, None)
    want = partial.double().reshape(G, bpg, 2, C).sum(1)
    assert torch.allclose(sums, want, rtol=1e-12, atol=1e-9)


def test_stem_pool_bn_bwd_fused_equals_unfused(backend):
    """the fused stem backward (no full-resolution gradient tensor) must reproduce
    maxpool_relu_bwd -> bn_bwd_reduce -> bn_bwd_apply"""
    lib = backend.hostlib
    g = torch.Generator().manual_seed(7)
    N, C, H, W, G = 4, 64, 12, 16, 2
    x = rb(torch.randn(N, C, H, W, generator=g))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    bnp, mpg = bn_forward_chain(lib, nhwc(x), gamma, beta, G, torch.zeros(C), torch.ones(C))
    Hp, Wp = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(N, Hp, Wp, C, dtype=torch.bfloat16)
    idx = torch.empty(N, Hp, Wp, C, dtype=torch.uint8)
    xpool = torch.empty(N, Hp, Wp, C, dtype=torch.bfloat16)
    lib.bn_relu_maxpool(nhwc(x), bnp, y, idx, xpool, N, H, W, C, Hp, Wp, N // G, None)
    gp = nhwc(rb(torch.randn(N, C, Hp, Wp, generator=g)))
    scratch = torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64)   # ticket counters + chunk sums
    M = N * H * W
    # unfused reference chain
    ga = torch.empty(N, H, W, C, dtype=torch.bfloat16)
    lib.maxpool_relu_bwd(gp, y, idx, ga, N, H, W, C, Hp, Wp, None)
    ppb = 32
    partial = torch.zeros(M // ppb, 2, C)
    lib.bn_bwd_reduce(ga, None, nhwc(x), bnp, partial, M, C, mpg, ppb, 0, None)
    sums = torch.zeros(G, 2, C, dtype=torch.float64)
    lib.bn_reduce_partials(partial, sums, scratch, G, (M // ppb) // G, C, None)
    dx = torch.empty(N, H, W, C, dtype=torch.bfloat16)
    lib.bn_bwd_apply(ga, None, nhwc(x), bnp, sums, dx, None, M, C, mpg, float(mpg), 0, None)
    # fused
    P = N * Hp * Wp
    ppb2 = 24
    partial2 = torch.zeros(P // ppb2, 2, C)
    lib.stem_pool_bn_bwd_reduce(gp, y, idx, nhwc(x), None, bnp, partial2, N, H, W, C, Hp, Wp, N // G, ppb2, None)
    partial3 = torch.zeros_like(partial2)
    lib.stem_pool_bn_bwd_reduce(gp, y, idx, nhwc(x), xpool, bnp, partial3, N, H, W, C, Hp, Wp, N // G, ppb2, None)
    assert torch.allclose(partial3, partial2, rtol=1e-5, atol=1e-4)    # pooled-raw stream == gather from raw
    sums2 = torch.zeros(G, 2, C, dtype=torch.float64)
    lib.bn_reduce_partials(partial2, sums2, scratch, G, (P // ppb2) // G, C, None)
    # the fused pass sums the per-window contributions before any bf16 rounding of the accumulated
    # gradient (the unfused chain rounds it when it materialises ga): equal to bf16 rounding noise
    assert torch.allclose(sums2, sums, rtol=1e-2, atol=0.1)
    dx2 = torch.empty(N, H, W, C, dtype=torch.bfloat16)
    lib.stem_pool_bn_bwd_apply(gp, y, idx, nhwc(x), bnp, sums, dx2, N, H, W, C, Hp, Wp, N // G, float(mpg), None)
    assert torch.equal(dx2, dx)


def test_stem_wgrad_fused_equals_unfused_chain(backend):
    """stem weight gradient with the BN-backward apply folded into the operand load vs the
    materialising chain (stem_pool_bn_bwd_apply -> stem_wgrad) on the same inputs"""
    from vfs_amd.packing import build_pack_table, wgrad_splits
    lib = backend.hostlib
    g = torch.Generator().manual_seed(8)
    N, H, W, G = 4, 40, 36, 2                       # input frames; stem output 20x18 (ragged 8x16 tiles)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
    img = rb(torch.randn(N, 3, H, W, generator=g))
    x4 = torch.zeros(N, H, W, 4, dtype=torch.bfloat16)
    lib.imgs_to_nhwc4(img.reshape(N, 1, 3, 1, H, W).contiguous(), x4, N, 1, 1, H, W, W, None)
    raw = rb(torch.randn(N, 64, Ho, Wo, generator=g))     # any raw conv output works for this identity
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    bnp, mpg = bn_forward_chain(lib, nhwc(raw), gamma, beta, G, torch.zeros(64), torch.ones(64))
    y = torch.empty(N, Hp, Wp, 64, dtype=torch.bfloat16)
    idx = torch.empty(N, Hp, Wp, 64, dtype=torch.uint8)
    lib.bn_relu_maxpool(nhwc(raw), bnp, y, idx, None, N, Ho, Wo, 64, Hp, Wp, N // G, None)
    gp = nhwc(rb(torch.randn(N, 64, Hp, Wp, generator=g)))
    P = N * Hp * Wp
    ppb = P // G
    partial = torch.zeros(P // ppb, 2, 64)
    lib.stem_pool_bn_bwd_reduce(gp, y, idx, nhwc(raw), None, bnp, partial, N, Ho, Wo, 64, Hp, Wp, N // G, ppb, None)
    sums = torch.zeros(G, 2, 64, dtype=torch.float64)
    lib.bn_reduce_partials(partial, sums, None, G, (P // ppb) // G, 64, None)
    count = float(mpg)
    # materialising chain
    dx = torch.empty(N, Ho, Wo, 64, dtype=torch.bfloat16)
    lib.stem_pool_bn_bwd_apply(gp, y, idx, nhwc(raw), bnp, sums, dx, N, Ho, Wo, 64, Hp, Wp, N // G, count, None)
    M = N * Ho * Wo
    nsplit, pps = wgrad_splits(M, 64, 256, target_blocks=6)
    part1 = torch.zeros(nsplit, 64, 256)
    grad1 = torch.zeros(64, 3, 7, 7)
    lib.stem_wgrad(dx, x4, part1, grad1, N, H, W, Ho, Wo, nsplit, pps, None)
    # fused
    ntiles = N * ((Ho + 7) // 8) * ((Wo + 15) // 16)
    tpb = (ntiles + 4) // 5
    nblocks = (ntiles + tpb - 1) // tpb
    part2 = torch.zeros(nblocks, 64, 224)
    grad2 = torch.ones(64, 3, 7, 7)
    lib.stem_wgrad_fused(x4, nhwc(raw), gp, y, idx, bnp, sums, part2, grad2, N, H, W, Ho, Wo, Hp, Wp, N // G, count,
                         nblocks, None)
    assert relerr(grad2 - 1.0, grad1) < 2e-4


@pytest.mark.parametrize('G,rows,C', [(2, 32, 2048), (2, 8, 96), (1, 200, 64)])
def test_bn_stats_from_raw_small_groups(backend, G, rows, C):
    """vfs_bn_stats_raw_finalize (the head's BN1d layers): sums, bnp = {scale, shift, mean, invstd} and the running
    statistics updated group by group, against torch on the same bf16 values; and vfs_loss_means"""
    lib = backend.hostlib
    g = torch.Generator().manual_seed(G * 100 + rows)
    x = rb(torch.randn(G * rows, C, generator=g) * 1.5 + 0.3)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    sums = torch.zeros(G, 2, C, dtype=torch.float64)
    bnp = torch.zeros(G, 4, C)
    lib.bn_stats_raw_finalize(x.to(torch.bfloat16), sums, gamma, beta, bnp, rm, rv, G, rows, C, float(rows), 1e-5, 0.1, None)
    for gi in range(G):
        xs = x[gi * rows:(gi + 1) * rows].double()
        assert torch.allclose(sums[gi, 0], xs.sum(0), rtol=1e-12, atol=1e-9)
        assert torch.allclose(sums[gi, 1], (xs * xs).sum(0), rtol=1e-12, atol=1e-9)
        mean, var = xs.mean(0), xs.var(0, unbiased=False)
        inv = 1.0 / torch.sqrt(var + 1e-5)
        assert torch.allclose(bnp[gi, 0].double(), gamma.double() * inv, rtol=2e-6)
        assert torch.allclose(bnp[gi, 1].double(), beta.double() - mean * gamma.double() * inv, rtol=1e-5, atol=1e-5)
        assert torch.allclose(bnp[gi, 2].double(), mean, rtol=1e-6, atol=1e-6) and torch.allclose(bnp[gi, 3].double(), inv, rtol=2e-6)
        rm0 = 0.9 * rm0 + 0.1 * mean.float()
        rv0 = 0.9 * rv0 + 0.1 * (var * rows / max(rows - 1, 1)).float()
    assert torch.allclose(rm, rm0, rtol=1e-5, atol=1e-6) and torch.allclose(rv, rv0, rtol=1e-5, atol=1e-6)
    loss = torch.rand(G + 1, rows, generator=g)
    means = torch.zeros(G + 2)
    lib.loss_means(loss, means, G + 1, rows, None)
    want = loss.double().mean(1)
    assert torch.allclose(means[:-1].double(), want, rtol=1e-6) and abs(float(means[-1]) - float(want.float().double().sum())) < 1e-6


@pytest.mark.parametrize('G,rows,C,ppr', [(2, 6, 128, 32), (1, 33, 64, 16), (2, 3, 32, 64), (2, 16, 256, 128)])
def test_bn_apply_with_inkernel_finalisation(backend, G, rows, C, ppr):
    """vfs_bn_act_fin == vfs_bn_stats_finalize + vfs_bn_act and vfs_bn_bwd_apply_fin == vfs_bn_bwd_sums_paramgrad +
    vfs_bn_bwd_apply on the same partial rows (bnp / sums / running statistics / dgamma / dbeta included)."""
    lib = backend.hostlib
    g = torch.Generator().manual_seed(rows * 7 + C)
    mpg = rows * ppr
    M = G * mpg
    x = rb(torch.randn(M, C, generator=g) * 1.3 + 0.2)
    xb = x.to(torch.bfloat16)
    part = torch.stack([x.view(G * rows, ppr, C).sum(1), (x * x).view(G * rows, ppr, C).sum(1)], dim=1).contiguous()   # [G*rows][2][C]
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    res = rb(torch.randn(M, C, generator=g)).to(torch.bfloat16)

    def fwd(fused):
        rm, rv = torch.full((C,), 0.25), torch.full((C,), 1.5)
        sums, bnp = torch.zeros(G, 2, C, dtype=torch.float64), torch.zeros(G, 4, C)
        y = torch.empty(M, C, dtype=torch.bfloat16)
        if fused:
            lib.bn_act_fin(xb, part, rows, gamma, beta, bnp, sums, rm, rv, res, None, None, y, M, C, mpg, 1, float(mpg), 1e-5, 0.1, None)
        else:
            scratch = torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64)
            lib.bn_stats_finalize(part, sums, scratch, gamma, beta, bnp, rm, rv, G, rows, C, float(mpg), 1e-5, 0.1, None)
            lib.bn_act(xb, bnp, res, None, None, y, M, C, mpg, 1, None)
        return y.float(), bnp, sums, rm, rv
    ya, bnpa, sa, rma, rva = fwd(True)
    yb, bnpb, sb, rmb, rvb = fwd(False)
    assert torch.allclose(sa, sb, rtol=1e-13, atol=1e-10) and torch.allclose(bnpa, bnpb, rtol=1e-6, atol=1e-7)
    assert torch.allclose(rma, rmb, rtol=1e-6, atol=1e-7) and torch.allclose(rva, rvb, rtol=1e-6, atol=1e-7)
    assert relerr(ya, yb) < 1e-6 or (ya - yb).abs().max() <= 2 ** -7 * yb.abs().max()

    # backward: partial rows of {sum g, sum g*xhat}; here any rows do - both paths consume the same ones
    gy = rb(torch.randn(M, C, generator=g)).to(torch.bfloat16)
    bpart = (torch.randn(G * rows, 2, C, generator=g) * 3).contiguous()

    def bwd(fused):
        bs = torch.zeros(G, 2, C, dtype=torch.float64)
        dg, db = torch.full((C,), 0.5), torch.full((C,), -0.25)       # gradients accumulate
        dx, gm = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
        if fused:
            lib.bn_bwd_apply_fin(gy, None, xb, bnpb, bpart, rows, bs, dg, db, dx, gm, M, C, mpg, float(mpg), 1, None)
        else:
            scratch = torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64)
            lib.bn_bwd_sums_paramgrad(bpart, bs, scratch, dg, db, G, rows, C, None)
            lib.bn_bwd_apply(gy, None, xb, bnpb, bs, dx, gm, M, C, mpg, float(mpg), 1, None)
        return dx.float(), gm.float(), bs, dg, db
    da, ga, bsa, dga, dba = bwd(True)
    dbb, gb, bsb, dgb, dbb2 = bwd(False)
    assert torch.allclose(bsa, bsb, rtol=1e-13, atol=1e-10)
    assert torch.allclose(dga, dgb, rtol=1e-6, atol=1e-6) and torch.allclose(dba, dbb2, rtol=1e-6, atol=1e-6)
    assert torch.equal(ga, gb)
    assert (da - dbb).abs().max() <= 2 ** -7 * dbb.abs().max()


@pytest.mark.parametrize('G,rows,C,ppr', [(2, 6, 128, 32), (1, 33, 64, 16), (2, 3, 32, 64), (2, 40, 256, 16)])
def test_bit_packed_relu_mask(backend, G, rows, C, ppr):
    """vfs_bn_act_mask / vfs_bn_act_fin_mask write y AND the bit-packed mask y > 0 (layout: tests/emu_util.pack_relu_mask);
    vfs_bn_bwd_reduce / _apply / _apply_fin with relu = 2 read that mask in place of y:
    every output is BIT-identical to the 16-byte mask operand."""
    lib = backend.hostlib
    g = torch.Generator().manual_seed(rows * 11 + C)
    mpg = rows * ppr
    M = G * mpg
    x = rb(torch.randn(M, C, generator=g) * 1.3 - 0.1)
    xb = x.to(torch.bfloat16)
    part = torch.stack([x.view(G * rows, ppr, C).sum(1), (x * x).view(G * rows, ppr, C).sum(1)], dim=1).contiguous()
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    res = rb(torch.randn(M, C, generator=g)).to(torch.bfloat16)
    rm, rv = torch.full((C,), 0.25), torch.full((C,), 1.5)
    sums, bnp = torch.zeros(G, 2, C, dtype=torch.float64), torch.zeros(G, 4, C)
    y0 = torch.empty(M, C, dtype=torch.bfloat16)
    lib.bn_act_fin(xb, part, rows, gamma, beta, bnp, sums, rm.clone(), rv.clone(), res, None, None, y0, M, C, mpg, 1, float(mpg), 1e-5, 0.1, None)
    want = pack_relu_mask(y0)
    assert 0.2 < (y0.float() > 0).float().mean() < 0.8
    # the two writers
    y1, m1 = torch.empty(M, C, dtype=torch.bfloat16), torch.full((M * C // 8,), 0xAA, dtype=torch.uint8)
    lib.bn_act_fin_mask(xb, part, rows, gamma, beta, torch.zeros(G, 4, C), torch.zeros(G, 2, C, dtype=torch.float64), rm.clone(), rv.clone(),
                        res, None, None, y1, m1, M, C, mpg, 1, float(mpg), 1e-5, 0.1, None)
    assert torch.equal(y1, y0) and torch.equal(m1, want)
    y2, m2 = torch.empty(M, C, dtype=torch.bfloat16), torch.full((M * C // 8,), 0x55, dtype=torch.uint8)
    lib.bn_act_mask(xb, bnp, res, None, None, y2, m2, M, C, mpg, 1, None)
    assert torch.equal(y2, y0) and torch.equal(m2, want)
    # without ReLU the mask still is y > 0 (negative values, zeros)
    y3, m3 = torch.empty(M, C, dtype=torch.bfloat16), torch.zeros(M * C // 8, dtype=torch.uint8)
    lib.bn_act_mask(xb, bnp, None, None, None, y3, m3, M, C, mpg, 0, None)
    assert torch.equal(m3, pack_relu_mask(y3))
    # the three readers
    gy = rb(torch.randn(M, C, generator=g)).to(torch.bfloat16)
    ppb = ppr
    pa, pb = torch.full((M // ppb, 2, C), float('nan')), torch.full((M // ppb, 2, C), float('nan'))
    lib.bn_bwd_reduce(gy, y0, xb, bnp, pa, M, C, mpg, ppb, 0, None)
    lib.bn_bwd_reduce(gy, m1, xb, bnp, pb, M, C, mpg, ppb, 2, None)
    assert torch.equal(pa, pb)
    bs = torch.zeros(G, 2, C, dtype=torch.float64)
    lib.bn_reduce_partials(pa, bs, torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64), G, (M // ppb) // G, C, None)
    outs = []
    for ym, rl in ((y0, 0), (m1, 2)):
        dx, gm = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
        lib.bn_bwd_apply(gy, ym, xb, bnp, bs, dx, gm, M, C, mpg, float(mpg), rl, None)
        dx2, gm2 = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
        bs2, dg, db = torch.zeros(G, 2, C, dtype=torch.float64), torch.zeros(C), torch.zeros(C)
        lib.bn_bwd_apply_fin(gy, ym, xb, bnp, pa, (M // ppb) // G, bs2, dg, db, dx2, gm2, M, C, mpg, float(mpg), rl, None)
        outs.append((dx, gm, dx2, gm2, bs2, dg, db))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][1].float(), torch.where(y0.float() > 0, gy.float(), torch.zeros(())))


def _bn_sweep(n, seed):
    import random
    rnd = random.Random(seed)
    out = []
    while len(out) < n:
        G = rnd.choice([1, 2])
        npg = rnd.choice([1, 2, 3])
        case = (G, npg, rnd.choice([3, 5, 7, 8, 9, 14]), rnd.choice([4, 7, 8, 13, 16]), rnd.choice([8, 16, 32, 64, 128, 192]))
        if case not in out:
            out.append(case)
    return out


@pytest.mark.parametrize('G,npg,H,W,C', _bn_sweep(14, 7))
def test_bn_shape_sweep(backend, G, npg, H, W, C):
    """BatchNorm apply / backward over odd pixel counts and every supported channel width (slab geometry of bn.hip):
    statistics from raw, bn_act (+ReLU), bn_bwd_reduce / apply with the mask recomputed from x, vs autograd per group"""
    lib = backend.hostlib
    g = torch.Generator().manual_seed(G * 1000 + H * 10 + C)
    N = G * npg
    M, mpg = N * H * W, npg * H * W
    x = rb(torch.randn(N, C, H, W, generator=g) * 1.7 + 0.4)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    sums, bnp = torch.zeros(G, 2, C, dtype=torch.float64), torch.zeros(G, 4, C)
    rm, rv = torch.zeros(C), torch.ones(C)
    xh = nhwc(x).to(torch.bfloat16)
    lib.bn_stats_raw_finalize(xh, sums, gamma, beta, bnp, rm, rv, G, mpg, C, float(mpg), 1e-5, 0.1, None)
    y = torch.empty(N, H, W, C, dtype=torch.bfloat16)
    lib.bn_act(xh, bnp, None, None, None, y, M, C, mpg, 1, None)
    xs = x.clone().requires_grad_(True)
    gm_, bt_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    outs = [F.relu(F.batch_norm(xs[i * npg:(i + 1) * npg], None, None, gm_, bt_, True, 0.1, 1e-5)) for i in range(G)]
    ref = torch.cat(outs)
    assert relerr(nchw(y), ref.detach()) < 6e-3
    gout = rb(torch.randn(N, C, H, W, generator=g))
    ref.backward(gout * (nchw(y) > 0))
    import math
    ppb = math.gcd(mpg, 512)
    if ppb < 16:
        ppb = mpg
    nblk = M // ppb
    partial = torch.zeros(nblk, 2, C)
    gh = nhwc(gout).to(torch.bfloat16)
    lib.bn_bwd_reduce(gh, y, xh, bnp, partial, M, C, mpg, ppb, 0, None)
    bs = torch.zeros(G, 2, C, dtype=torch.float64)
    dg, db = torch.zeros(C), torch.zeros(C)
    dx = torch.empty(N, H, W, C, dtype=torch.bfloat16)
    lib.bn_bwd_apply_fin(gh, y, xh, bnp, partial, nblk // G, bs, dg, db, dx, None, M, C, mpg, float(mpg), 0, None)
    assert relerr(nchw(dx), xs.grad) < 1e-2
    assert relerr(dg, gm_.grad) < 2e-3 and relerr(db, bt_.grad) < 2e-3


@pytest.mark.parametrize('G,mpg,C', [(2, 100, 128), (2, 72, 256), (1, 50, 512), (2, 20, 1024), (1, 9, 2048)])
def test_wide_slab_streaming_equals_64_channel_slabs(backend, G, mpg, C):
    """round 3: the plain (non-FIN) bn_act / bn_bwd_apply launches on >= 128-channel tensors stream whole pixel rows per workgroup
    (`bn_wide`, up to 512 channels per slab) instead of 64-channel slabs - every output, the bit-packed mask included, must be
    BIT-identical to the 64-channel-slab launch (ragged row counts, both mask modes, residual operands)"""
    lib = backend.hostlib
    g = torch.Generator().manual_seed(C + mpg)
    M = G * mpg
    x = rb(torch.randn(M, C, generator=g) * 1.3 - 0.1).to(torch.bfloat16)
    res = rb(torch.randn(M, C, generator=g)).to(torch.bfloat16)
    rres = rb(torch.randn(M, C, generator=g)).to(torch.bfloat16)
    gy = rb(torch.randn(M, C, generator=g)).to(torch.bfloat16)
    bnp = torch.randn(G, 4, C, generator=g) * 0.5 + 0.8
    rbnp = torch.randn(G, 4, C, generator=g) * 0.5 + 0.8
    bs = torch.randn(G, 2, C, generator=g, dtype=torch.float64)

    def run(wide):
        backend.lib.set_option(b'bn_wide', wide)
        backend.lib.set_option(b'bn_wide_min_mb', 0)
        out = []
        for r, rr, relu in ((res, None, 1), (None, rres, 1), (None, None, 0)):
            y, mb = torch.empty(M, C, dtype=torch.bfloat16), torch.zeros(M * C // 8, dtype=torch.uint8)
            lib.bn_act_mask(x, bnp, r, rr, rbnp if rr is not None else None, y, mb, M, C, mpg, relu, None)
            out += [y, mb]
        y0, m0 = out[0], out[1]
        for ym, rl in ((y0, 0), (m0, 2), (None, 1)):
            dx, gm = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
            lib.bn_bwd_apply(gy, ym, x, bnp, bs, dx, gm, M, C, mpg, float(mpg), rl, None)
            out += [dx, gm]
        return out
    try:
        narrow, wide = run(0), run(1)
    finally:
        backend.lib.set_option(b'bn_wide', 1)
        backend.lib.set_option(b'bn_wide_min_mb', 8)
    for a, b in zip(narrow, wide):
        assert torch.equal(a, b)
    assert torch.equal(narrow[1], pack_relu_mask(narrow[0]))


@pytest.mark.gpu
def test_ticketed_reduction_under_memory_pressure(gpu_backend):
    """the chunked single-launch reduction - chunk sums handed to the last-ticket workgroup through agent-scope stores - repeated
    400 times on 8192 statistics rows while a second stream keeps the memory system busy; every repetition must reproduce the first
    one bit for bit and equal the fp64 host sum.  (Written after the round-3 memory-ordering fix in vfs_release_workgroup; the race
    itself - ~1 % of full ResNet-18 eager steps with the pre-fix library, MEASUREMENTS.md - does NOT reproduce in this isolated loop,
    so this is a determinism check of the protocol, not a reproducer.)"""
    lib, dev = gpu_backend.lib, gpu_backend.dev
    G, bpg, C = 2, 4096, 64
    g = torch.Generator().manual_seed(5)
    part = torch.randn(G * bpg, 2, C, generator=g).to(dev)
    want = part.double().view(G, bpg, 2, C).sum(1).cpu()
    scratch = torch.zeros(32 + G * 128 * 2 * C, dtype=torch.float64, device=dev)
    big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    big_b = torch.empty_like(big_a)
    side = torch.cuda.Stream(dev)
    outs = []
    for it in range(400):
        if it % 4 == 0:
            with torch.cuda.stream(side):
                big_b.copy_(big_a)
        sums = torch.zeros(G, 2, C, dtype=torch.float64, device=dev)
        lib.bn_reduce_partials(part, sums, scratch, G, bpg, C, None)
        outs.append(sums)
    torch.cuda.synchronize()
    first = outs[0].cpu()
    assert torch.allclose(first, want, rtol=1e-12, atol=1e-9)
    bad = [i for i, o in enumerate(outs) if not torch.equal(o.cpu(), first)]
    assert not bad, bad[:10]


@pytest.mark.parametrize('G,mpg,K,C,relu', [(2, 32, 2048, 64, 1), (2, 32, 512, 2048, 0), (2, 16, 256, 128, 1), (1, 40, 128, 32, 1), (4, 8, 128, 16, 0),
                                            (2, 64, 256, 64, 1), (2, 128, 512, 128, 1), (1, 200, 128, 32, 0)])
def test_linear_bn_act_one_launch_equals_three(backend, G, mpg, K, C, relu):
    """round 6: vfs_linear_bn_act (nn.Linear + BatchNorm1d + ReLU of the SimSiam head in one launch, sim_siam_head.py:78-111) against the
    three launches it replaces - vfs_conv_fwd (skinny GEMM), vfs_bn_stats_raw_finalize, vfs_bn_act: raw, act, bnp, sums and the
    running statistics bit for bit (same GEMM order, statistics of the stored values in the same summation order)"""
    lib = backend.hostlib
    g = torch.Generator().manual_seed(G * 1000 + mpg * 10 + C)
    M = G * mpg
    x = rb(torch.randn(M, K, generator=g)).to(torch.bfloat16)
    w = rb(torch.randn(C, K, generator=g) * (1.0 / K) ** 0.5).to(torch.bfloat16)
    bias = torch.randn(C, generator=g) * 0.1
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3

    def three():
        raw = torch.empty(M, C, dtype=torch.bfloat16)
        lib.conv_fwd(x.view(M, 1, 1, K), w.view(C, 1, 1, K), raw.view(M, 1, 1, C), bias, None, M, 1, 1, K, 1, 1, C, 1, 1, 1, 0, None)
        sums, bnp = torch.zeros(G, 2, C, dtype=torch.float64), torch.zeros(G, 4, C)
        rm, rv = torch.full((C,), 0.25), torch.full((C,), 1.5)
        lib.bn_stats_raw_finalize(raw, sums, gamma, beta, bnp, rm, rv, G, mpg, C, float(mpg), 1e-5, 0.1, None)
        act = torch.empty(M, C, dtype=torch.bfloat16)
        lib.bn_act(raw, bnp, None, None, None, act, M, C, mpg, relu, None)
        return raw, act, bnp, sums, rm, rv

    def one():
        raw, act = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
        sums, bnp = torch.zeros(G, 2, C, dtype=torch.float64), torch.zeros(G, 4, C)
        rm, rv = torch.full((C,), 0.25), torch.full((C,), 1.5)
        lib.linear_bn_act(x, w, bias, gamma, beta, raw, act, bnp, sums, rm, rv, M, K, C, mpg, relu, float(mpg), 1e-5, 0.1, None)
        return raw, act, bnp, sums, rm, rv
    a, b = three(), one()
    names = ('raw', 'act', 'bnp', 'sums', 'running_mean', 'running_var')
    for n, ta, tb in zip(names, a, b):
        if M > 128:      # above 128 rows vfs_conv_fwd is the implicit-GEMM kernel: another K order, raw equal to one bf16 ulp
            assert relerr(tb.double(), ta.double()) < (6e-3 if ta.dtype == torch.bfloat16 else 2e-3), n
        elif ta.dtype == torch.bfloat16:
            assert torch.equal(ta.view(torch.int16), tb.view(torch.int16)), n
        else:
            assert torch.equal(ta, tb), n
    if M > 128:      # ... and the statistics are those of ITS OWN stored values, exactly
        sums, bnp = torch.zeros(G, 2, C, dtype=torch.float64), torch.zeros(G, 4, C)
        rm, rv = torch.full((C,), 0.25), torch.full((C,), 1.5)
        lib.bn_stats_raw_finalize(b[0], sums, gamma, beta, bnp, rm, rv, G, mpg, C, float(mpg), 1e-5, 0.1, None)
        act = torch.empty(M, C, dtype=torch.bfloat16)
        lib.bn_act(b[0], bnp, None, None, None, act, M, C, mpg, relu, None)
        assert torch.equal(sums, b[3]) and torch.equal(bnp, b[2]) and torch.equal(rm, b[4]) and torch.equal(rv, b[5])
        assert torch.equal(act.view(torch.int16), b[1].view(torch.int16))
    ref = torch.nn.functional.linear(x.float(), w.float(), bias)
    assert relerr(b[0].float(), ref) < 6e-3
    with pytest.raises(Exception):
        lib.linear_bn_act(x, w, bias, gamma, beta, b[0], b[1], b[2], b[3], None, None, M, K, C, mpg + 1, relu, float(mpg), 1e-5, 0.1, None)


@pytest.mark.parametrize('G,mpg,C,rl', [(2, 32, 2048, 1), (2, 32, 512, 0), (1, 64, 64, 1), (2, 24, 32, 1), (3, 8, 128, 2), (2, 256, 64, 1)])
def test_bn_bwd_apply_raw_one_launch_equals_two(backend, G, mpg, C, rl):
    """round 6: vfs_bn_bwd_apply_raw (groups with ONE statistics row: the SimSiam head's BatchNorm1d layers) against vfs_bn_bwd_reduce +
    vfs_bn_bwd_apply_fin on the same operands - sums, dgamma, dbeta, dx, gm bit for bit (relu modes: none, recomputed from x, bit mask)"""
    import math
    lib = backend.hostlib
    g = torch.Generator().manual_seed(G * 100 + mpg + C)
    M = G * mpg
    x = rb(torch.randn(M, C, generator=g) * 1.3 + 0.2).to(torch.bfloat16)
    gy = rb(torch.randn(M, C, generator=g)).to(torch.bfloat16)
    bnp = torch.stack([torch.rand(G, C, generator=g) + 0.5, torch.randn(G, C, generator=g) * 0.4, torch.randn(G, C, generator=g) * 0.2,
                       torch.rand(G, C, generator=g) + 0.7], 1).contiguous()
    ym = None
    if rl == 2:
        from tests.emu_util import pack_relu_mask
        ym = pack_relu_mask(torch.randn(M, C, generator=g))
    ppb = math.gcd(mpg, 512)
    if ppb < 16:
        ppb = mpg
    assert M // ppb == G

    def run(one):
        bs = torch.zeros(G, 2, C, dtype=torch.float64)
        dg, db = torch.full((C,), 0.5), torch.full((C,), -0.25)
        dx, gm = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
        if one:
            lib.bn_bwd_apply_raw(gy, ym, x, bnp, bs, dg, db, dx, gm, M, C, mpg, float(mpg), rl, None)
        else:
            part = torch.zeros(G, 2, C)
            lib.bn_bwd_reduce(gy, ym, x, bnp, part, M, C, mpg, ppb, rl, None)
            lib.bn_bwd_apply_fin(gy, ym, x, bnp, part, 1, bs, dg, db, dx, gm, M, C, mpg, float(mpg), rl, None)
        return bs, dg, db, dx, gm
    a, b = run(False), run(True)
    for n, ta, tb in zip(('sums', 'dgamma', 'dbeta', 'dx', 'gm'), a, b):
        if ta.dtype == torch.bfloat16:
            assert torch.equal(ta.view(torch.int16), tb.view(torch.int16)), n
        else:
            assert torch.equal(ta, tb), n
    with pytest.raises(Exception):      # three rows per group: not this entry point's shape
        lib.bn_bwd_apply_raw(gy[:96], None, x[:96], bnp, a[0], a[1], a[2], a[3][:96], None, 96, C, 48, 48.0, 0, None)
