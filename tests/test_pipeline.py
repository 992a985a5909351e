"""Training input pipeline: host decision sampling vs decisions captured from the
reference's own RandomResizedCrop / Flip (tests/golden/pipeline_decisions.npz), and the fused
crop+resize+flip+normalise kernel vs oracle/pipeline_oracle.py, bit-exact.  backend=emu (CPU) / gpu."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as PO
from vfs_amd.pipeline import GpuTrainPipeline

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pipeline_decisions.npz'))
CASES = sorted({k.split('/')[0] for k in GOLD.files})
MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def _cfg(case, out=224):
    soc, sac = (bool(v) for v in GOLD[case + '/same'])
    return [dict(type='RandomResizedCrop', area_range=tuple(GOLD[case + '/area_range']), same_across_clip=sac, same_on_clip=soc),
            dict(type='Resize', scale=(out, out), keep_ratio=False),
            dict(type='Flip', flip_ratio=float(GOLD[case + '/flip_ratio']), same_across_clip=sac, same_on_clip=soc),
            dict(type='Normalize', mean=MEAN, std=STD, to_bgr=False),
            dict(type='FormatShape', input_format='NCTHW'),
            dict(type='Collect', keys=['imgs', 'label'], meta_keys=[]),
            dict(type='ToTensor', keys=['imgs', 'label'])]


@pytest.mark.parametrize('case', CASES)
def test_decisions_match_reference(case):
    hs, ws, nclips, clip_len, nsamp, seed = (int(v) for v in GOLD[case + '/meta'])
    soc, sac = (bool(v) for v in GOLD[case + '/same'])
    nf = nclips * clip_len
    # the oracle's restatement
    np.random.seed(seed)
    random.seed(seed)
    boxes, flips = [], []
    for _ in range(nsamp):
        boxes.append(PO.sample_crops(nf, clip_len, (hs, ws), tuple(GOLD[case + '/area_range']), same_on_clip=soc, same_across_clip=sac))
        flips.append(PO.sample_flips(nf, clip_len, float(GOLD[case + '/flip_ratio']), same_on_clip=soc, same_across_clip=sac))
    assert np.array_equal(np.concatenate(boxes), GOLD[case + '/boxes'])
    assert np.array_equal(np.concatenate(flips), GOLD[case + '/flips'])
    # the product's host sampling
    np.random.seed(seed)
    random.seed(seed)
    pipe = GpuTrainPipeline(_cfg(case), nclips, clip_len)
    got = [pipe.sample(nf, (hs, ws)) for _ in range(nsamp)]
    assert np.array_equal(np.concatenate([g[0] for g in got]), GOLD[case + '/boxes'])
    assert np.array_equal(np.concatenate([g[1] for g in got]), GOLD[case + '/flips'])


def test_reference_config_parses():
    from vfs_amd.config import Config
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfgs = [os.path.join(here, 'configs', n) for n in os.listdir(os.path.join(here, 'configs'))] if os.path.isdir(os.path.join(here, 'configs')) else []
    cfgs = [c for c in cfgs if c.endswith('.py')]
    if not cfgs:
        pytest.skip('no config files in the repo')
    for c in cfgs:
        cfg = Config.fromfile(c)
        if 'train_pipeline' not in cfg:
            continue
        pipe = GpuTrainPipeline(cfg.train_pipeline)
        assert pipe.out_hw == (224, 224) and pipe.crop is not None and pipe.flip is not None
        assert (pipe.num_clips, pipe.clip_len) == ((2, 4) if 'r18' in c else (2, 1))


def test_oracle_resize_known_answers():
    """arithmetic identities of cv2's fixed-point bilinear: same-size resize is the identity, constants stay,
    2x upsampling of a two-pixel ramp gives the quarter points rounded to nearest"""
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    assert np.array_equal(PO.resize_bilinear_u8(img, 17, 13), img)
    const = np.full((9, 11, 3), 201, np.uint8)
    assert np.array_equal(PO.resize_bilinear_u8(const, 40, 23), np.full((23, 40, 3), 201, np.uint8))
    ramp = np.array([[[0], [255]]], np.uint8)
    assert PO.resize_bilinear_u8(ramp, 4, 1)[0, :, 0].tolist() == [0, 64, 191, 255]
    col = ramp.transpose(1, 0, 2)
    assert PO.resize_bilinear_u8(col, 1, 4)[:, 0, 0].tolist() == [0, 64, 191, 255]
    n = PO.normalize(np.array([[[0, 128, 255]]], np.uint8), MEAN, STD)
    want = (np.array([0, 128, 255], np.float64) - np.array(MEAN)) / np.array(STD)
    assert np.allclose(n[0, 0], want, rtol=0, atol=2e-7)


def _frames(B, F, Hs, Ws, seed):
    g = np.random.default_rng(seed)
    base = g.integers(0, 256, (B, F, Hs // 4 + 1, Ws // 4 + 1, 3), dtype=np.uint8)
    up = np.repeat(np.repeat(base, 4, axis=2), 4, axis=3)[:, :, :Hs, :Ws]
    noise = g.integers(-20, 21, up.shape)
    return np.clip(up.astype(np.int64) + noise, 0, 255).astype(np.uint8)


def run_pipeline_case(be, B, V, T, Hs, Ws, Ho, Wo, seed, boxes=None, flips=None):
    frames = _frames(B, V * T, Hs, Ws, seed)
    pipe = GpuTrainPipeline([dict(type='RandomResizedCrop', area_range=(0.2, 1.), same_across_clip=False, same_on_clip=False),
                             dict(type='Resize', scale=(Wo, Ho), keep_ratio=False),
                             dict(type='Flip', flip_ratio=0.5, same_across_clip=False, same_on_clip=False),
                             dict(type='Normalize', mean=MEAN, std=STD, to_bgr=False),
                             dict(type='FormatShape', input_format='NCTHW')], V, T)
    np.random.seed(seed)
    random.seed(seed)
    from vfs_amd import _lib
    prev = _lib._LIB
    _lib.set_lib(be.lib)
    try:
        out = pipe(be.d(torch.from_numpy(frames)), boxes=boxes, flips=flips, want_x4=True)
        if be.dev.type == 'cuda':
            torch.cuda.synchronize()
    finally:
        _lib.set_lib(prev)
    want = PO.train_pipeline(frames, out['boxes'], out['flips'], (Ho, Wo), MEAN, STD, V, T)
    got = out['imgs'].cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.abs(got - want).max())
    # the bf16 NHWC4 output = vfs_imgs_to_nhwc4 of the fp32 output (frame order (v, b, t), zero pad channel / column)
    Wp = Wo + (Wo & 1)
    x4 = torch.full((V * B * T, Ho, Wp, 4), 7.0, dtype=torch.bfloat16)
    be.hostlib.imgs_to_nhwc4(torch.from_numpy(want), x4, B, V, T, Ho, Wo, Wp, None)
    assert torch.equal(out['x4'].cpu().view(torch.int16), x4.view(torch.int16))
    return out


@pytest.mark.parametrize('shape', [(2, 2, 1, 64, 80, 32, 32), (1, 2, 2, 40, 56, 24, 17), (1, 1, 1, 20, 20, 48, 48)])
def test_kernel_matches_oracle(backend, shape):
    out = run_pipeline_case(backend, *shape, seed=3)
    assert out['flips'].shape[0] == shape[0] * shape[1] * shape[2]


def test_kernel_edge_boxes(backend):
    """full-frame, one-pixel-wide, one-pixel-high and corner boxes; every frame flipped"""
    B, V, T, Hs, Ws = 1, 2, 2, 33, 47
    boxes = np.array([[0, 0, Ws, Hs], [5, 3, 6, 30], [2, 9, 40, 10], [Ws - 3, Hs - 2, Ws, Hs]], np.int32)
    run_pipeline_case(backend, B, V, T, Hs, Ws, 16, 20, seed=5, boxes=boxes, flips=np.ones(4, np.uint8))
    run_pipeline_case(backend, B, V, T, Hs, Ws, 16, 20, seed=6, boxes=boxes, flips=np.zeros(4, np.uint8))


@pytest.mark.gpu
def test_kernel_full_size(gpu_backend):
    """BASELINE size: 340x256 decoded frames -> 224x224, 8 pairs; checked against the oracle on the GPU box"""
    run_pipeline_case(gpu_backend, 8, 2, 1, 256, 340, 224, 224, seed=9)


@pytest.mark.gpu
def test_pipeline_feeds_train_step(gpu_backend):
    """decoded uint8 frames -> GpuTrainPipeline (built from the shipped config's train_pipeline) -> train_step: the imgs
    tensor equals the oracle pipeline's bit for bit, so the step's log vars equal those of the oracle-fed step"""
    import vfs_amd
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = vfs_amd.Config.fromfile(os.path.join(here, 'configs', 'vfs_r18.py'))
    tp = [dict(s) for s in cfg.train_pipeline]
    for s in tp:                                  # a small crop keeps the test quick; everything else as shipped
        if s['type'] == 'Resize':
            s['scale'] = (64, 64)
    pipe = GpuTrainPipeline(tp)
    V, T = pipe.num_clips, pipe.clip_len
    assert (V, T) == (2, 4)
    B = 4
    frames = _frames(B, V * T, 72, 96, 21)
    np.random.seed(1)
    random.seed(1)
    out = pipe(torch.from_numpy(frames).to(gpu_backend.dev))
    want = PO.train_pipeline(frames, out['boxes'], out['flips'], (64, 64), pipe.mean, pipe.std, V, T)
    assert np.array_equal(out['imgs'].cpu().numpy().view(np.uint32), want.view(np.uint32))
    model = vfs_amd.build_model(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(gpu_backend.dev).train()
    res = model.train_step(dict(imgs=out['imgs'], label=torch.zeros(B, 1)), None)
    res['loss'].backward()
    assert np.isfinite(res['log_vars']['loss']) and res['num_samples'] == B
    res2 = model.train_step(dict(imgs=torch.from_numpy(want).to(gpu_backend.dev), label=torch.zeros(B, 1)), None)
    assert res2['log_vars'].keys() == res['log_vars'].keys()


def test_pipeline_config_errors_and_clip_layout():
    from vfs_amd.pipeline import clips_from_pipeline
    base = [dict(type='SampleFrames', clip_len=1, num_clips=8), dict(type='Clip2Frame', clip_len=4),
            dict(type='Resize', scale=(32, 24), keep_ratio=False), dict(type='Normalize', mean=MEAN, std=STD, to_bgr=False)]
    assert clips_from_pipeline(base) == (2, 4)
    assert clips_from_pipeline([dict(type='SampleFrames', clip_len=3, num_clips=2)]) == (2, 3)
    with pytest.raises(ValueError):
        clips_from_pipeline([dict(type='Resize', scale=(8, 8))])
    pipe = GpuTrainPipeline(base)
    assert pipe.out_hw == (24, 32) and pipe.crop is None and pipe.flip is None
    boxes, flips = pipe.sample(8, (50, 70))          # no crop / flip steps: full frames, never flipped
    assert boxes.tolist() == [[0, 0, 70, 50]] * 8 and not flips.any()
    for bad in (dict(type='ColorJitter'), dict(type='Resize', scale=(8, 8)), dict(type='Flip', direction='vertical'),
                dict(type='Normalize', mean=MEAN, std=STD, to_bgr=True), dict(type='FormatShape', input_format='NCHW')):
        with pytest.raises(NotImplementedError):
            GpuTrainPipeline(base + [bad])
    with pytest.raises(ValueError):
        GpuTrainPipeline([dict(type='SampleFrames', clip_len=1, num_clips=2)])     # no Resize / Normalize
