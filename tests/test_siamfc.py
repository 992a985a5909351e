"""SiamFC probe (§8f rank 4): heads.  Oracle pinned against responses of the reference classes
(tests/golden/siamfc_heads.npz); the HIP heads (vfs_xcorr_fwd + 1x1 vfs_conv_fwd) against the oracle on the same
bf16-rounded operands.  backend=emu (CPU) / gpu."""
import os

import numpy as np
import pytest
import torch

from oracle import siamfc_oracle as SO
from oracle import vfs_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'siamfc_heads.npz'))
TAGS = sorted({k.split('/')[0] for k in G.files})


def _inputs(tag):
    nz, nx, c, hz, h = (int(v) for v in G[tag + '/shape'])
    return nz, nx, c, O.fill_tensor([nz, c, hz, hz], 3, scale=1.5), O.fill_tensor([nx, c, h, h + 2], 4, scale=1.5)


@pytest.mark.parametrize('tag', TAGS)
def test_oracle_heads_match_reference(tag):
    nz, nx, c, z, x = _inputs(tag)
    with torch.no_grad():
        got = SO.SiamFC(out_scale=0.001)(z, x).numpy()
        head = SO.SiamConvFC(c, 2 * c, out_scale=0.01)
        assert list(head.state_dict().keys()) == [str(k) for k in G[tag + '/keys']]
        O.fill_state_dict_(head, seed=21)
        got2 = head(z, x).numpy()
    assert np.allclose(got, G[tag + '/siamfc'], rtol=1e-5, atol=1e-7)
    assert np.allclose(got2, G[tag + '/siamconvfc'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('tag', TAGS)
def test_hip_heads_match_oracle(backend, tag):
    import vfs_amd
    nz, nx, c, z, x = _inputs(tag)
    zr, xr = O.round_bf16(z), O.round_bf16(x)
    dev = backend.dev
    with torch.no_grad():
        want = SO.SiamFC(out_scale=0.001)(zr, xr)
        got = vfs_amd.SiamFC(out_scale=0.001)(zr.to(dev), xr.to(dev)).cpu()
        assert got.shape == want.shape
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-9      # fp32 sums of identical bf16 products
        ref = SO.SiamConvFC(c, 2 * c, out_scale=0.01)
        O.fill_state_dict_(ref, seed=21)
        head = vfs_amd.SiamConvFC(c, 2 * c, out_scale=0.01)
        head.load_state_dict(ref.state_dict())
        head.to(dev)
        got2 = head(zr.to(dev), xr.to(dev)).cpu()
        want2 = ref(zr, xr)
        # bf16 weights and bf16 conv outputs in the HIP head: one rounding of each 1x1 conv
        assert float((got2 - want2).norm() / want2.norm()) < 1.5e-2
    with pytest.raises(NotImplementedError):
        vfs_amd.SiamConvFC(c, c, kernel_size=3)


# ---------------------------------------------------------------------------------------------
# training the probe: goldens produced by the reference's own losses / heads / _create_labels / torch optimizers
# (tests/golden/gen_siamfc_train_golden.py)
# ---------------------------------------------------------------------------------------------
GT = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'siamfc_train.npz'))
TRAIN_CASES = {'focal_adam': ('focal', 'Adam', 1e-3, 0.0), 'balance_sgd': ('balance', 'SGD', 1e-2, 5e-4), 'balance_adam_wd': ('balance', 'Adam', 1e-3, 5e-4)}


def _train_inputs():
    return O.fill_tensor([4, 64, 5, 5], 3, scale=1.5), O.fill_tensor([4, 64, 12, 12], 4, scale=1.5)


@pytest.mark.parametrize('tag', sorted(TRAIN_CASES))
def test_oracle_training_matches_reference(tag):
    loss_name, optname, lr, wd = TRAIN_CASES[tag]
    zf, xf = _train_inputs()
    head = SO.SiamConvFC(64, 64, out_scale=0.01)
    O.fill_state_dict_(head, seed=21)
    resp = head(zf, xf)
    labels = SO.create_labels(resp.size(), 16, 0, 8)
    assert np.array_equal(labels.numpy(), GT[tag + '/labels'])
    assert np.allclose(resp.detach().numpy(), GT[tag + '/responses'], rtol=1e-5, atol=1e-7)
    loss = SO.focal_loss(resp, labels) if loss_name == 'focal' else SO.balanced_loss(resp, labels)
    assert abs(float(loss) - float(GT[tag + '/loss'])) < 1e-6 * max(1.0, abs(float(GT[tag + '/loss'])))
    loss.backward()
    for n, p in head.named_parameters():
        g = GT[f'{tag}/grad/{n}']
        assert np.allclose(p.grad.numpy(), g, rtol=1e-4, atol=1e-6 * np.abs(g).max()), n


def test_oracle_losses_with_soft_ring_match_reference():
    lab = SO.create_labels((2, 1, 9, 9), 16, 32, 8)
    assert np.array_equal(lab.numpy(), GT['ring/labels']) and set(np.unique(lab.numpy())) == {0.0, 0.5, 1.0}
    for tag, fn in (('ring/balance', lambda x: SO.balanced_loss(x, lab, 0.5)), ('ring/focal', lambda x: SO.focal_loss(x, lab, 1.5))):
        x = O.fill_tensor([2, 1, 9, 9], 7, scale=3.0).requires_grad_(True)
        loss = fn(x)
        loss.backward()
        assert abs(float(loss) - float(GT[tag + '/loss'])) < 1e-6
        assert np.allclose(x.grad.numpy(), GT[tag + '/grad'], rtol=1e-4, atol=1e-7)


def test_hip_loss_kernels_match_reference(backend):
    lib = backend.hostlib
    lab = torch.from_numpy(GT['ring/labels'])
    x = O.fill_tensor([2, 1, 9, 9], 7, scale=3.0)
    for tag, mode, param in (('ring/balance', 0, 0.5), ('ring/focal', 1, 1.5)):
        loss, grad = torch.zeros(1), torch.zeros_like(x)
        lib.siamfc_loss(x, lab, loss, grad, x.numel(), mode, param, 1.0, None)
        assert abs(float(loss) - float(GT[tag + '/loss'])) < 2e-6 * max(1.0, abs(float(GT[tag + '/loss'])))
        g = GT[tag + '/grad']
        assert np.allclose(grad.numpy(), g, rtol=2e-4, atol=2e-7 + 1e-5 * np.abs(g).max()), tag


def test_hip_xcorr_backward_matches_autograd(backend):
    lib = backend.hostlib
    g = torch.Generator().manual_seed(3)
    for nz, nx, c, hz, h, w in ((2, 2, 64, 3, 7, 8), (2, 6, 16, 2, 5, 5)):
        z = O.round_bf16(torch.randn(nz, c, hz, hz, generator=g)).requires_grad_(True)
        x = O.round_bf16(torch.randn(nx, c, h, w, generator=g)).requires_grad_(True)
        out = SO.fast_xcorr(z, x) * 0.05
        go = torch.randn(out.shape, generator=g)
        out.backward(go)
        zb, xb = z.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16), x.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
        dz, dx = torch.zeros_like(zb), torch.zeros_like(xb)
        lib.xcorr_bwd(zb, xb, go.contiguous(), dz, dx, nz, nx, hz, hz, h, w, c, 0.05, None)
        wz, wx = z.grad.permute(0, 2, 3, 1), x.grad.permute(0, 2, 3, 1)
        assert float((dz.float() - wz).abs().max()) <= 2 ** -8 * float(wz.abs().max()) * 1.05
        assert float((dx.float() - wx).abs().max()) <= 2 ** -8 * float(wx.abs().max()) * 1.05


@pytest.mark.parametrize('tag', sorted(TRAIN_CASES))
def test_hip_probe_training_matches_reference(backend, tag):
    """one training iteration of the probe's head on given backbone features: loss, parameter gradients and two optimizer
    steps against what the reference's classes + torch.optim produced (bf16 operands in the HIP head)"""
    import vfs_amd
    from vfs_amd import siamfc as SF
    loss_name, optname, lr, wd = TRAIN_CASES[tag]
    dev = backend.dev
    zf, xf = _train_inputs()
    head = vfs_amd.SiamConvFC(64, 64, out_scale=0.01)
    ref = SO.SiamConvFC(64, 64, out_scale=0.01)
    O.fill_state_dict_(ref, seed=21)
    head.load_state_dict(ref.state_dict())
    head.to(dev)
    params = list(head.parameters())
    opt = SF.Adam(params, lr=lr, weight_decay=wd) if optname == 'Adam' else SF.ParamSGD(params, lr=lr, weight_decay=wd, momentum=0.9)
    labels = SF.create_labels((4, 1, 8, 8), 16, 0, 8, dev)
    assert np.array_equal(labels.cpu().numpy(), GT[tag + '/labels'])
    for step in range(2):
        for p in params:
            p.grad = torch.zeros_like(p)
        loss, resp = SF.head_loss_backward(head, zf.to(dev), xf.to(dev), labels, loss_name)
        if step == 0:
            want = GT[tag + '/responses']
            assert float(np.abs(resp.cpu().numpy() - want).max()) < 2e-2 * float(np.abs(want).max())
            assert abs(float(loss) - float(GT[tag + '/loss'])) < 2e-2 * abs(float(GT[tag + '/loss']))
            for n, p in head.named_parameters():
                g = torch.from_numpy(GT[f'{tag}/grad/{n}'])
                rel = float((p.grad.cpu() - g).norm() / g.norm())
                assert rel < 3e-2, (n, rel)
        opt.step()
        for n, p in head.named_parameters():
            w = torch.from_numpy(GT[f'{tag}/step{step + 1}/{n}'])
            # Adam's first steps move every weight by ~lr regardless of the gradient's size: compare the UPDATE
            w0 = ref.state_dict()[n]
            upd, want_upd = p.detach().cpu() - w0, w - w0
            assert float((upd - want_upd).norm() / want_upd.norm()) < (0.12 if optname == 'Adam' else 3e-2), (n, step)


def test_head_uses_weights_after_optimizer_steps(backend):
    """the probe's optimizers write p.data through raw pointers (no tensor._version bump): the head's packed bf16 weight copies
    must refresh anyway - responses after N large Adam steps equal those of a FRESH head holding the updated weights"""
    import vfs_amd
    from vfs_amd import siamfc as SF
    dev = backend.dev
    zf, xf = _train_inputs()
    head = vfs_amd.SiamConvFC(64, 64, out_scale=0.01)
    ref = SO.SiamConvFC(64, 64, out_scale=0.01)
    O.fill_state_dict_(ref, seed=21)
    head.load_state_dict(ref.state_dict())
    head.to(dev)
    params = list(head.parameters())
    opt = SF.Adam(params, lr=0.1)
    labels = SF.create_labels((4, 1, 8, 8), 16, 0, 8, dev)
    first = None
    for step in range(3):
        for p in params:
            p.grad = torch.zeros_like(p)
        _, resp = SF.head_loss_backward(head, zf.to(dev), xf.to(dev), labels, 'balance')
        first = resp.cpu().clone() if first is None else first
        opt.step()
    with torch.no_grad():
        got = head(zf.to(dev), xf.to(dev)).cpu()
        fresh = vfs_amd.SiamConvFC(64, 64, out_scale=0.01)
        fresh.load_state_dict(head.state_dict())
        want = fresh.to(dev)(zf.to(dev), xf.to(dev)).cpu()
    assert torch.equal(got, want)
    assert float((got - first).abs().max()) > 0.05 * float(first.abs().max())      # the weights really moved


def test_adam_kernel_equals_torch(backend):
    lib = backend.hostlib
    g = torch.Generator().manual_seed(0)
    p = torch.randn(1000, generator=g)
    tp = p.clone().requires_grad_(True)
    topt = torch.optim.Adam([tp], lr=1e-3, weight_decay=1e-2)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        tp.grad = gr.clone()
        topt.step()
        lib.adam_step(p, gr, m, v, 1000, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, None)
        assert float((p - tp.detach()).abs().max()) < 1e-6


def test_probe_train_step_and_tracking_loop(backend):
    """SiamFCProbe end to end on the device: a training iteration through the frozen dilated backbone lowers nothing it
    should not (backbone untouched, head updated, finite loss), labels as the reference builds them, and the tracking loop
    follows a bright square that moves a few pixels per frame (crops / cubic up-sampling are restated cv2: unpinned)"""
    import vfs_amd
    if backend.name == 'emu':
        pytest.skip('full ResNet-18 at 120 / 255 pixel crops: minutes on the emulator; runs on the GPU')
    cfg = dict(out_channels=512, batch_size=2, exemplar_sz=120, instance_sz=255)
    probe = vfs_amd.SiamFCProbe(cfg, depth=18, device=backend.dev)
    ref = O.ResNet(18, strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4), out_indices=(3,), zero_init_residual=False)
    O.fill_state_dict_(ref, seed=118)
    probe.backbone.load_state_dict(ref.state_dict())
    before = {k: v.clone() for k, v in probe.backbone.state_dict().items()}
    hb = {k: v.clone() for k, v in probe.head.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    z = torch.rand(2, 3, 120, 120, generator=g) * 255
    x = torch.rand(2, 3, 255, 255, generator=g) * 255
    l1 = probe.train_step((z, x))
    l2 = probe.train_step((z, x), backward=False)
    assert np.isfinite(l1) and np.isfinite(l2)
    for k, v in probe.backbone.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert any(not torch.equal(v, hb[k]) for k, v in probe.head.state_dict().items())
    assert tuple(probe.labels.shape) == (2, 1, 18, 18) and float(probe.labels.max()) == 1.0
    # tracking loop
    frames = []
    for t in range(4):
        img = np.full((240, 320, 3), 60, np.uint8)
        img[100 + 3 * t:140 + 3 * t, 150 + 4 * t:190 + 4 * t] = 220
        frames.append(img)
    boxes = probe.track(frames, [151, 101, 40, 40])
    assert boxes.shape == (4, 4) and np.isfinite(boxes).all()
