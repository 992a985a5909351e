"""Pins oracle/exact_oracle.c (the bit-defined C restatement of the reference's fp32 evaluation path) against
vectors captured from the reference itself (tests/golden/gen_*.py ran the reference's own classes):
features to fp32 summation-order accuracy, label maps pixel for pixel."""
import os

import numpy as np
import pytest
import torch

from oracle import exact_oracle as X
from oracle import vfs_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(REPO, 'tests', 'golden')


def test_exp_polynomial_is_accurate():
    x = -np.abs(np.random.RandomState(0).randn(2000).astype(np.float32)) * 20
    got = X.exp_le0(x).astype(np.float64)
    want = np.exp(x.astype(np.float64))
    ok = want > 1e-37
    assert np.max(np.abs(got[ok] - want[ok]) / want[ok]) < 3e-7
    assert X.exp_le0(np.float32(0.0)) == np.float32(1.0)


@pytest.mark.parametrize('depth', [18, 50])
def test_resnet_eval_matches_reference_dilated(depth):
    """the reference class in eval mode (SiamFC probe settings: dilations (1,1,2,4), strides (1,2,1,1))"""
    g = np.load(os.path.join(G, f'resnet{depth}_dilated_eval.npz'))
    net = O.ResNet(depth, strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4), out_indices=(3,), zero_init_residual=False)
    O.fill_state_dict_(net, seed=depth + 100)
    x = O.fill_tensor([2, 3, 64, 80], seed=9, scale=2.0)
    y = X.resnet_eval(net.state_dict(), depth, x, strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4), out_indices=(3,))[3]
    y = np.transpose(y, (0, 3, 1, 2))
    assert tuple(y.shape) == tuple(g['shape'])
    flat = y.reshape(-1)
    want = g['sample']
    rel = np.abs(flat[::13] - want).max() / np.abs(want).max()
    assert rel < 1e-5, rel
    assert abs(flat.astype(np.float64).sum() - g['checksum'][0]) < 1e-5 * g['checksum'][1]


def test_masked_attention_matches_reference():
    """masked_attention_efficient of the reference on a 12x16 map, 5 key frames, radius 4, top-10"""
    g = np.load(os.path.join(G, 'masked_attention.npz'))
    q = O.fill_tensor([1, 16, 12, 16], 21).numpy()
    k = O.fill_tensor([1, 16, 5, 12, 16], 22).numpy()
    v = np.abs(O.fill_tensor([1, 3, 5, 12, 16], 23).numpy())
    H, W = 12, 16
    # bank = the 5 key frames followed by the query frame, rows [frame][pixel][channel]
    feats = np.concatenate([np.transpose(k[0], (1, 2, 3, 0)), np.transpose(q[0], (1, 2, 0))[None]])     # [6,H,W,C]
    bank = X.l2norm_rows(feats.reshape(-1, 16)).reshape(6, H * W, 16)
    sbank = np.zeros((6, H * W, 3), np.float32)
    sbank[:5] = np.transpose(v[0], (1, 2, 3, 0)).reshape(5, H * W, 3)
    for radius, key in ((4, 'out'), (0, 'out_nomask')):
        out = X.labelprop(bank, sbank, 5, [0, 1, 2, 3, 4], H, W, radius, 10, 0.07)
        want = np.transpose(g[key][0], (1, 2, 0)).reshape(H * W, 3)
        assert np.abs(out - want).max() < 2e-5 * max(1.0, np.abs(want).max()), key


def _davis_model(all_blocks=False):
    import vfs_amd
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', 'vfs_r18.py'))
    tc = dict(cfg.test_cfg)
    tc['neighbor_range'] = 8
    tc['precede_frames'] = 3
    if all_blocks:
        tc['all_blocks'] = True
    ref = O.VanillaTracker(18, tc)
    O.fill_state_dict_(ref, seed=5)
    return ref, tc


def _davis_clip():
    T, H, W = 6, 96, 128
    imgs = O.fill_tensor([1, 1, 3, T, H, W], 41, scale=2.0)
    yy, xx = np.mgrid[0:H, 0:W]
    seg = np.zeros((H, W), np.uint8)
    seg[(yy > 20) & (yy < 60) & (xx > 30) & (xx < 70)] = 1
    seg[(yy > 50) & (yy < 90) & (xx > 80) & (xx < 120)] = 2
    return imgs, seg, (H, W)


def test_forward_test_matches_reference_labels():
    """VanillaTracker.forward_test of the reference (R18 test-time config, 6-frame 96x128 clip): every pixel of
    every propagated frame, and the res4 features the reference extracted"""
    g = np.load(os.path.join(G, 'forward_test_r18.npz'))
    ref, tc = _davis_model()
    imgs, seg, hw = _davis_clip()
    assert np.array_equal(seg, g['ref_seg'])
    sd = ref.state_dict()
    frames = imgs[0, 0].permute(1, 0, 2, 3).numpy()
    feat = X.resnet_eval(sd, 18, frames, strides=(1, 2, 1, 1), out_indices=(2,), prefix='backbone.')[2]
    nchw = np.transpose(feat, (0, 3, 1, 2))
    assert tuple(nchw.shape) == tuple(g['feat_shape'])
    flat = nchw.reshape(-1)
    assert np.abs(flat[::997] - g['feat_sample']).max() < 1e-5 * np.abs(g['feat_sample']).max()
    out = X.forward_test(sd, 18, imgs, seg, hw + (3,), tc)
    agree = float((out == g['seg_preds']).mean())
    assert agree == 1.0, f'label agreement with the reference {agree:.6f}'


def test_forward_test_all_blocks_matches_reference_labels():
    g = np.load(os.path.join(G, 'forward_test_r18_all_blocks.npz'))
    ref, tc = _davis_model(all_blocks=True)
    imgs, seg, hw = _davis_clip()
    out = X.forward_test(ref.state_dict(), 18, imgs, seg, hw + (3,), tc)
    assert out.shape == g['seg_preds'].shape
    agree = float((out == g['seg_preds']).mean())
    assert agree >= 0.9999, f'label agreement with the reference {agree:.6f}'
