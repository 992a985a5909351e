"""Pin the CPU oracle (oracle/vfs_oracle.py) against vectors captured from the
real reference by tests/golden/gen_golden.py.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import vfs_oracle as O

G = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(G, name + '.npz'), allow_pickle=False)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('depth', [18, 50])
def test_resnet_forward_matches_reference(depth):
    g = load(f'resnet{depth}_fwd')
    net = O.ResNet(depth, out_indices=(0, 1, 2, 3))
    assert list(net.state_dict().keys()) == [str(k) for k in g['keys']]
    O.fill_state_dict_(net, seed=depth)
    net.train()
    outs = net(O.fill_tensor([2, 3, 64, 64], seed=7, scale=2.0))
    for i in range(4):
        assert rel(outs[i].detach().numpy(), g[f'out{i}']) < 1e-5, i
    sd = net.state_dict()
    assert rel(sd['conv1.bn.running_mean'].numpy(), g['stem_running_mean']) < 1e-6
    assert rel(sd['conv1.bn.running_var'].numpy(), g['stem_running_var']) < 1e-6


def test_cosine_loss_matches_reference():
    g = load('cosine_loss')
    p, z = O.fill_tensor([6, 32], 1), O.fill_tensor([6, 32], 2)
    assert rel(O.cosine_sim_loss(p, z).numpy(), g['loss']) < 1e-6
    assert rel(O.cosine_sim_loss(p, z, negative=True).numpy(), g['loss_neg']) < 1e-6


@pytest.mark.parametrize('tag,depth,shape', [('r18', 18, [2, 2, 3, 4, 64, 64]),
                                             ('r50', 50, [4, 2, 3, 1, 64, 64])])
def test_train_step_matches_reference(tag, depth, shape):
    g = load(f'{tag}_train')
    model = O.build_tracker(depth)
    assert list(model.state_dict().keys()) == [str(k) for k in g['keys']]
    # init statistics of the real reference vs ours (kaiming fan_out / BN 1,0 /
    # zero-init residual gamma / torch Linear default) -- same distribution family
    for k, v in model.state_dict().items():
        key = 'init/' + k
        if key in g.files and (k.endswith('bn.weight') or k.endswith('bn.bias')):
            assert abs(float(v.float().mean()) - g[key][0]) < 1e-6, k
    O.fill_state_dict_(model, seed=3)
    model.train()
    imgs = O.fill_tensor(shape, seed=11, scale=2.0)
    losses = model.forward_train(imgs)
    loss, log_vars = O.parse_losses(losses)
    loss.backward()
    assert abs(float(loss) - float(g['loss'])) < 1e-5 * max(1, abs(float(g['loss'])))
    for k, v in losses.items():
        assert rel(v.detach().numpy(), g['lossvec/' + k]) < 2e-5, k
        assert abs(log_vars[k] - float(g['log/' + k])) < 1e-5
    bad = []
    for n, p in model.named_parameters():
        gn = float(g['gnorm/' + n])
        mine = float(p.grad.double().norm())
        if abs(mine - gn) > 2e-3 * max(gn, 1e-6) + 1e-7:
            bad.append((n, mine, gn))
        s = p.grad.flatten()[:: max(1, p.grad.numel() // 16)][:16].numpy()
        ref = g['gsample/' + n]
        assert np.abs(s - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-6) + 1e-7, n
    assert not bad, bad[:5]
    # SGD step (configs/*:134)
    params = [p for _, p in model.named_parameters()]
    before = [p.detach().clone() for p in params]
    with torch.no_grad():
        O.sgd_step(params, [p.grad for p in params], [None] * len(params), lr=0.05)
    for (n, p), b in zip(model.named_parameters(), before):
        d = (p.detach() - b).flatten()
        d = d[:: max(1, d.numel() // 8)][:8].numpy()
        ref = g['delta/' + n]
        assert np.abs(d - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-9) + 1e-9, n


def test_spatial_neighbor_matches_reference():
    g = load('spatial_neighbor')
    m = O.spatial_neighbor_circle(12, 16, 8).numpy()
    ref = np.unpackbits(g['mask'])[: m.size].reshape(m.shape).astype(bool)
    assert (m == ref).all()


def test_masked_attention_matches_reference():
    g = load('masked_attention')
    q, k = O.fill_tensor([1, 16, 12, 16], 21), O.fill_tensor([1, 16, 5, 12, 16], 22)
    v = O.fill_tensor([1, 3, 5, 12, 16], 23).abs()
    m = O.spatial_neighbor_circle(12, 16, 8)
    o = O.masked_attention_efficient(q, k, v, m, 0.07, 10, True)
    assert rel(o.numpy(), g['out']) < 1e-5
    o2 = O.masked_attention_efficient(q, k, v, None, 0.07, 10, True)
    assert rel(o2.numpy(), g['out_nomask']) < 1e-5


def test_pil_nearest_matches_reference():
    g = load('pil_nearest')
    lab = (O.fill_tensor([480, 854], 31).numpy() * 2.5 + 2.5).astype(np.uint8)
    assert (O.pil_nearest_resize(lab, 60, 107) == g['out']).all()


def test_forward_test_matches_reference():
    g = load('forward_test_r18')
    tc = dict(precede_frames=3, topk=10, temperature=0.07, strides=(1, 2, 1, 1), out_indices=(2,),
              neighbor_range=8, with_first=True, with_first_neighbor=True)
    model = O.VanillaTracker(18, tc)
    O.fill_state_dict_(model, seed=5)
    model.eval()
    T, H, W = 6, 96, 128
    imgs = O.fill_tensor([1, 1, 3, T, H, W], 41, scale=2.0)
    seg = g['ref_seg']
    out = model.forward_test(imgs, seg, (H, W, 3))
    assert out.shape == g['seg_preds'].shape and out.dtype == np.uint8
    mism = (out != g['seg_preds']).mean()
    assert mism == 0.0, mism


def test_forward_test_all_blocks_matches_reference():
    """test_cfg.all_blocks=True (vanilla_tracker.py:30-46, README.md:76): one label map per res4 block"""
    g = load('forward_test_r18_all_blocks')
    tc = dict(precede_frames=3, topk=10, temperature=0.07, strides=(1, 2, 1, 1), out_indices=(2,),
              neighbor_range=8, with_first=True, with_first_neighbor=True, all_blocks=True)
    model = O.VanillaTracker(18, tc)
    O.fill_state_dict_(model, seed=5)
    model.eval()
    T, H, W = 6, 96, 128
    imgs = O.fill_tensor([1, 1, 3, T, H, W], 41, scale=2.0)
    out = model.forward_test(imgs, g['ref_seg'], (H, W, 3))
    assert out.shape == g['seg_preds'].shape == (2, T, H, W) and out.dtype == np.uint8
    assert (out != g['seg_preds']).mean() == 0.0



@pytest.mark.parametrize('depth', [18, 50])
def test_dilated_resnet_eval_matches_reference(depth):
    """the backbone as the SiamFC probe builds it (dilations (1,1,2,4), strides (1,2,1,1), eval): oracle vs features
    captured from the reference class (tests/golden/gen_dilated_golden.py)"""
    g = load(f'resnet{depth}_dilated_eval')
    net = O.ResNet(depth, strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4), out_indices=(3,), zero_init_residual=False)
    assert list(net.state_dict().keys()) == [str(k) for k in g['keys']]
    O.fill_state_dict_(net, seed=depth + 100)
    net.eval()
    with torch.no_grad():
        y = net(O.fill_tensor([2, 3, 64, 80], seed=9, scale=2.0))
    assert list(y.shape) == list(g['shape'])
    flat = y.flatten()
    assert rel(flat[::13].numpy(), g['sample']) < 1e-5
    assert abs(flat.double().abs().sum().item() - g['checksum'][1]) < 1e-5 * g['checksum'][1]


def test_resnet50_caffe_style_matches_reference():
    """style='caffe' (resnet.py:156-161): stage outputs of the reference class, train mode (tests/golden/gen_caffe_golden.py)"""
    g = load('resnet50_caffe_fwd')
    net = O.ResNet(50, out_indices=(0, 1, 2, 3), style='caffe')
    assert list(net.state_dict().keys()) == [str(k) for k in g['keys']]
    O.fill_state_dict_(net, seed=50)
    net.train()
    outs = net(O.fill_tensor([2, 3, 64, 64], seed=7, scale=2.0))
    for i, o in enumerate(outs):
        assert list(o.shape) == list(g[f'shape{i}'])
        assert rel(o.detach().flatten()[::7].numpy(), g[f'sample{i}']) < 1e-5, i
