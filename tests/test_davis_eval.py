"""DAVIS J&F evaluation: the oracle restatement against hand-computable cases,
the HIP count kernels (through the C ABI) bit-exact against the oracle, the host evaluator against the
oracle's metric dict, palette PNG round trip, checkpoint key conversion.

The reference's metric lives in the un-vendored davis2017 package: parity UNPINNED (oracle/davis_jf.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import davis_jf as O


def blobs(seed, T=5, H=40, W=56, nobj=3, jitter=2):
    """ground truth = moving discs / boxes, prediction = the same shapes jittered and eroded a little"""
    rng = np.random.RandomState(seed)
    gt = np.zeros((T, H, W), np.uint8)
    pred = np.zeros((T, H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(1, nobj + 1):
        cy, cx, r = rng.randint(8, H - 8), rng.randint(8, W - 8), rng.randint(4, 10)
        for t in range(T):
            cy2, cx2 = cy + t * rng.randint(-1, 2), cx + t * rng.randint(-1, 2)
            if k % 2:
                m = (yy - cy2) ** 2 + (xx - cx2) ** 2 <= r * r
            else:
                m = (abs(yy - cy2) <= r) & (abs(xx - cx2) <= r // 2 + 1)
            gt[t][m] = k
            dy, dx = rng.randint(-jitter, jitter + 1, 2)
            pm = np.roll(np.roll(m, dy, 0), dx, 1)
            if (t + k) % 4 == 0:
                pm = np.zeros_like(pm)          # a missed object: empty-boundary conventions
            pred[t][pm] = k
    return pred, gt


def test_oracle_hand_cases():
    # identical masks: J = F = 1; disjoint far apart: J = 0, F = 0
    m = np.zeros((20, 20), bool); m[5:12, 6:14] = True
    c = O.frame_counts(m, m)
    assert O.j_from_counts(*c[:2]) == 1.0 and O.f_from_counts(*c[2:]) == 1.0
    n = np.zeros((20, 20), bool); n[15:19, 0:3] = True
    c = O.frame_counts(m, n, bound_th=1)
    assert O.j_from_counts(*c[:2]) == 0.0 and O.f_from_counts(*c[2:]) == 0.0
    # empty vs empty: J = 1 (empty union), F = 1; empty prediction vs object: precision 1, recall 0 -> F = 0
    z = np.zeros((20, 20), bool)
    c = O.frame_counts(z, z)
    assert O.j_from_counts(*c[:2]) == 1.0 and O.f_from_counts(*c[2:]) == 1.0
    c = O.frame_counts(z, m)
    assert O.j_from_counts(*c[:2]) == 0.0 and O.f_from_counts(*c[2:]) == 0.0
    # seg2bmap of a 2x2 block in the middle: the 3x3 "north-west" ring (8 pixels incl. 3 of the block)
    s = np.zeros((6, 6), bool); s[2:4, 2:4] = True
    b = O.seg2bmap(s)
    assert b.sum() == 8 and b[1, 1] and b[3, 3] and not b[2, 2] and not b[4, 4]
    # boundary radius of a DAVIS frame, disk footprint
    assert O.bound_pixels(480, 854) == 8 and O.disk(2).sum() == 13
    # statistics: bins of the decay
    v = np.linspace(1.0, 0.0, 10)
    m_, r_, d_ = O.db_statistics(v)
    assert abs(m_ - 0.5) < 1e-12 and abs(r_ - 0.5) < 1e-12 and d_ > 0.6


@pytest.mark.parametrize('seed,shape,nobj', [(0, (5, 40, 56), 3), (1, (3, 33, 47), 1), (2, (6, 64, 64), 5), (3, (4, 17, 90), 2)])
def test_counts_kernel_bit_exact(backend, seed, shape, nobj):
    pred, gt = blobs(seed, *shape, nobj=nobj)
    if seed == 2:
        rows = np.nonzero(pred[1].any(1))[0]
        r0 = int(rows[len(rows) // 2])
        gt[:, r0:r0 + 3, :] = 255       # void band through predicted objects (ignored unless use_void)
    want = O.sequence_counts(pred, gt)
    T, H, W = shape
    K = want.shape[0]
    assert K == nobj
    lib, dev = backend.lib, backend.dev
    F = T - 2
    counts = torch.full((F, K, 6), -1, dtype=torch.int32, device=dev)
    scratch = torch.empty(2 * F * H * W, dtype=torch.int32, device=dev)
    lib.davis_counts(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev), counts, scratch, T, H, W, K,
                     O.bound_pixels(H, W), 0, None)
    got = counts.cpu().numpy().astype(np.int64).transpose(1, 0, 2)
    assert np.array_equal(got, want)
    if seed == 2:                       # void pixels honoured
        want_v = O.sequence_counts(pred, gt, use_void=True)
        lib.davis_counts(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev), counts, scratch, T, H, W, K,
                         O.bound_pixels(H, W), 1, None)
        assert np.array_equal(counts.cpu().numpy().astype(np.int64).transpose(1, 0, 2), want_v)
        assert not np.array_equal(want_v, want)


@pytest.mark.gpu
def test_counts_kernel_davis_size(gpu_backend):
    """a full-size DAVIS clip (480x854, boundary disk radius 8), three objects"""
    pred, gt = blobs(7, 6, 480, 854, nobj=3, jitter=6)
    want = O.sequence_counts(pred, gt)
    from vfs_amd import davis_eval as DE
    from vfs_amd._lib import set_lib
    set_lib(gpu_backend.lib)
    got = DE.sequence_counts(torch.from_numpy(pred).to(gpu_backend.dev), torch.from_numpy(gt).to(gpu_backend.dev))
    assert np.array_equal(got, want)


def test_evaluator_matches_oracle_dict(backend):
    import vfs_amd
    from vfs_amd import davis_eval as DE
    from vfs_amd._lib import set_lib, get_lib
    seqs = {f'seq{i}': blobs(10 + i, 7, 48, 64, nobj=1 + i % 3) for i in range(3)}
    want = O.evaluate(seqs)
    prev = None
    try:
        try:
            prev = get_lib()
        except Exception:
            prev = None
        set_lib(backend.lib)
        ev = vfs_amd.DavisEvaluator([g for _, g in seqs.values()], names=list(seqs), device=backend.dev)
        got = ev.evaluate([p for p, _ in seqs.values()], metrics='davis')
        with pytest.raises(KeyError):
            ev.evaluate([p for p, _ in seqs.values()], metrics='daivs')
    finally:
        set_lib(prev)
    for k in DE.G_MEASURES:
        assert abs(got[k] - want[k]) < 1e-12, (k, got[k], want[k])
    for name, (j, f) in want['per_object'].items():
        assert abs(ev.per_object[name][0] - j) < 1e-12 and abs(ev.per_object[name][1] - f) < 1e-12


def test_palette_png_round_trip(tmp_path):
    from vfs_amd import davis_eval as DE
    pred, _ = blobs(5, 4, 30, 44, nobj=3)
    DE.save_palette_pngs([pred], str(tmp_path), ['bike'])
    back = DE.load_palette_pngs(os.path.join(str(tmp_path), 'bike'))
    assert np.array_equal(back, pred)
    from PIL import Image
    img = Image.open(os.path.join(str(tmp_path), 'bike', '00001.png'))
    assert img.mode == 'P' and img.getpalette()[:9] == [0, 0, 0, 128, 0, 0, 0, 128, 0]


def test_checkpoint_key_conversion():
    import vfs_amd
    from vfs_amd.checkpoint import from_pretrained_keys, to_pretrained_keys
    for depth in (18, 50):
        net = vfs_amd.ResNet(depth=depth, out_indices=(3,))
        sd = {'backbone.' + k: v for k, v in net.state_dict().items()}
        sd['img_head.projection_fcs.0.weight'] = torch.zeros(1)      # dropped by the conversion
        tv = to_pretrained_keys(sd)
        assert 'conv1.weight' in tv and 'bn1.running_mean' in tv and 'layer1.0.conv1.weight' in tv
        assert 'layer2.0.downsample.0.weight' in tv and 'layer2.0.downsample.1.bias' in tv
        assert not any(k.startswith('img_head') or k.startswith('backbone') for k in tv)
        assert len(tv) == len(net.state_dict())
        # the torchvision-style dict loads back (resnet.py:488-523 mapping) and round-trips the names
        back = from_pretrained_keys(tv)
        assert set(back) == set(k for k in sd if k.startswith('backbone.'))
        net2 = vfs_amd.ResNet(depth=depth, out_indices=(3,))
        for p in net2.parameters():
            torch.nn.init.normal_(p)
        net2.load_torchvision_checkpoint({'state_dict': tv})
        for (k1, v1), (k2, v2) in zip(net.state_dict().items(), net2.state_dict().items()):
            assert k1 == k2 and torch.equal(v1, v2), k1
    with pytest.raises(RuntimeError):
        to_pretrained_keys({'backbone.layer1.0.conv1.foo.weight': torch.zeros(1)})
