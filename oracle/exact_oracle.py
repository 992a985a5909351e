"""TEST INFRASTRUCTURE -- Python face of oracle/exact_oracle.c (see its header): the reference's fp32
evaluation path (ResNet eval forward, masked_attention_efficient, post-processing) with every result
defined to the last bit.  The HIP "exact" kernels (vfs_amd/csrc/exact_f32.hip) must match it bit for
bit; it is itself pinned against vectors captured from the reference (tests/test_exact_oracle.py).

Only tests/, __graft_entry__ (build + smoke) and bench.py's cpu_baseline import this module."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'exact_oracle.c')
LIB = os.path.join(HERE, '_build', 'libexact_oracle.so')

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)
_U8 = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp: plain C, no dependencies"""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(['gcc', '-O3', '-mavx2', '-mfma', '-ffp-contract=off', '-fno-math-errno', '-fopenmp', '-shared',
                           '-fPIC', '-o', LIB, SRC, '-lm'])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.xo_exp.restype = ctypes.c_float
        _lib.xo_exp.argtypes = [ctypes.c_float]
    return _lib


def _f(a):
    assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(_F)


def f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def bn_eval_affine(gamma, beta, mean, var, eps):
    """BatchNorm in eval mode as one affine map, each step a single fp32 operation:
    scale = gamma / sqrt(var + eps), shift = beta - mean * scale."""
    gamma, beta, mean, var = (np.asarray(t, dtype=np.float32) for t in (gamma, beta, mean, var))
    scale = gamma / np.sqrt(var + np.float32(eps))
    shift = beta - mean * scale
    return f32(scale), f32(shift)


def conv2d(x, w_oihw, scale=None, shift=None, res=None, stride=1, pad=0, dil=1, relu=False):
    """x [N,H,W,Cin] fp32 NHWC, w [Cout,Cin,KH,KW] (the reference's layout) -> [N,Ho,Wo,Cout]"""
    x = f32(x)
    N, H, W, Cin = x.shape
    Cout, _, KH, KW = w_oihw.shape
    Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    w = f32(np.transpose(np.asarray(w_oihw, dtype=np.float32), (2, 3, 1, 0)))      # [KH][KW][Cin][Cout]
    y = np.empty((N, Ho, Wo, Cout), np.float32)
    sc = _f(f32(scale)) if scale is not None else None
    sh = _f(f32(shift)) if shift is not None else None
    if res is not None:
        res = f32(res)
        assert res.shape == y.shape
    lib().xo_conv2d(_f(x), _f(w), sc, sh, _f(res) if res is not None else None, _f(y), N, H, W, Cin, Ho, Wo, Cout, KH, KW,
                    stride, pad, dil, 1 if relu else 0)
    return y


def maxpool3x3s2(x):
    x = f32(x)
    N, H, W, C = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = np.empty((N, Ho, Wo, C), np.float32)
    lib().xo_maxpool3x3s2(_f(x), _f(y), N, H, W, C, Ho, Wo)
    return y


def l2norm_rows(x):
    x = f32(x)
    P, C = x.shape
    assert C % 4 == 0
    y = np.empty_like(x)
    lib().xo_l2norm_rows(_f(x), _f(y), ctypes.c_longlong(P), C)
    return y


def exp_le0(x):
    return np.array([lib().xo_exp(float(np.float32(v))) for v in np.ravel(x)], np.float32).reshape(np.shape(x))


def labelprop(fbank, sbank, qframe, kslot, H, W, radius, topk, temperature, non_mask_len=0):
    """fbank [frames, H*W, C] (L2-normalised), sbank [frames, H*W, CO] -> [H*W, CO]"""
    fbank, sbank = f32(fbank), f32(sbank)
    C, CO = fbank.shape[2], sbank.shape[2]
    ks = np.ascontiguousarray(np.asarray(kslot, np.int32))
    out = np.empty((H * W, CO), np.float32)
    lib().xo_labelprop(_f(fbank), _f(sbank), _f(out), int(qframe), ks.ctypes.data_as(_I), len(ks), H, W, C, CO, int(radius),
                       int(non_mask_len), int(topk), ctypes.c_float(temperature))
    return out


def seg_postprocess(seg, H, W, Ho, Wo):
    """seg [H*W, CO] fp32 -> uint8 [Ho, Wo]"""
    seg = f32(seg)
    CO = seg.shape[-1]
    lab = np.empty((Ho, Wo), np.uint8)
    lib().xo_seg_postprocess(_f(seg), lab.ctypes.data_as(_U8), H, W, CO, Ho, Wo)
    return lab


# ---------------------------------------------------------------------------------------------
# ResNet in eval mode from a state_dict with the reference's key names (resnet.py:555-575)
# ---------------------------------------------------------------------------------------------
ARCH = {18: ('basic', (2, 2, 2, 2)), 34: ('basic', (3, 4, 6, 3)), 50: ('bottleneck', (3, 4, 6, 3)),
        101: ('bottleneck', (3, 4, 23, 3)), 152: ('bottleneck', (3, 8, 36, 3))}


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)


def _unit(sd, name, x, stride, pad, dil, relu, res=None, eps=1e-5):
    """mmcv ConvModule in eval mode: conv(bias=False) -> BN(running statistics) -> [+ identity] -> [ReLU]"""
    scale, shift = bn_eval_affine(_np(sd[name + '.bn.weight']), _np(sd[name + '.bn.bias']), _np(sd[name + '.bn.running_mean']),
                                  _np(sd[name + '.bn.running_var']), eps)
    return conv2d(x, _np(sd[name + '.conv.weight']), scale, shift, res=res, stride=stride, pad=pad, dil=dil, relu=relu)


def resnet_eval(sd, depth, x_nchw, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(3,), prefix='', all_blocks=False):
    """x [N,3,H,W] fp32 -> {stage: NHWC fp32 output}; with all_blocks a list of every block output of the stages in
    out_indices (vanilla_tracker.py:30-46).  Stops after the last requested stage."""
    kind, nblocks = ARCH[depth]
    x = f32(np.transpose(_np(x_nchw), (0, 2, 3, 1)))
    x = _unit(sd, prefix + 'conv1', x, 2, 3, 1, True)
    x = maxpool3x3s2(x)
    outs, blocks = {}, []
    for si, nb in enumerate(nblocks):
        dil = dilations[si]
        for bi in range(nb):
            name = f'{prefix}layer{si + 1}.{bi}'
            stride = strides[si] if bi == 0 else 1
            d = (dil if dil == 1 else dil // 2) if bi == 0 else dil     # make_res_layer, resnet.py:279-300
            identity = x
            if name + '.downsample.conv.weight' in sd:
                identity = _unit(sd, name + '.downsample', x, stride, 0, 1, False)
            if kind == 'basic':
                o = _unit(sd, name + '.conv1', x, stride, d, d, True)
                x = _unit(sd, name + '.conv2', o, 1, 1, 1, True, res=identity)
            else:
                o = _unit(sd, name + '.conv1', x, 1, 0, 1, True)
                o = _unit(sd, name + '.conv2', o, stride, d, d, True)
                x = _unit(sd, name + '.conv3', o, 1, 0, 1, True, res=identity)
            if si in out_indices:
                blocks.append(x)
        if si in out_indices:
            outs[si] = x
        if si >= max(out_indices):
            break
    return blocks if all_blocks else outs


def pil_nearest_resize(label, out_h, out_w):
    """common/utils.py:25-42 (PIL NEAREST): src = floor((dst + 0.5) * in / out)"""
    in_h, in_w = label.shape
    ys = np.minimum(np.floor((np.arange(out_h) + 0.5) * (in_h / out_h)).astype(np.int64), in_h - 1)
    xs = np.minimum(np.floor((np.arange(out_w) + 0.5) * (in_w / out_w)).astype(np.int64), in_w - 1)
    return label[ys][:, xs]


def torch_nearest_resize(label, out_h, out_w):
    """F.interpolate(mode='nearest'): src = floor(dst * in / out) in fp32"""
    in_h, in_w = label.shape
    ys = np.minimum(np.floor(np.arange(out_h) * np.float32(in_h / out_h)).astype(np.int64), in_h - 1)
    xs = np.minimum(np.floor(np.arange(out_w) * np.float32(in_w / out_w)).astype(np.int64), in_w - 1)
    return label[ys][:, xs]


def bilinear_resize(src, Ho, Wo, src_nhwc, dst_nhwc):
    """F.interpolate(bilinear, align_corners=False): src [C,H,W] (or [H,W,C] when src_nhwc) -> [C,Ho,Wo] / [Ho,Wo,C]"""
    src = f32(src)
    if src_nhwc:
        H, W, C = src.shape
        ss = (1, W * C, C)
    else:
        C, H, W = src.shape
        ss = (H * W, W, 1)
    dst = np.empty((Ho, Wo, C) if dst_nhwc else (C, Ho, Wo), np.float32)
    ds = (1, Wo * C, C) if dst_nhwc else (Ho * Wo, Wo, 1)
    LL = ctypes.c_longlong
    lib().xo_bilinear_resize(_f(src), _f(dst), C, H, W, Ho, Wo, LL(ss[0]), LL(ss[1]), LL(ss[2]), LL(ds[0]), LL(ds[1]), LL(ds[2]))
    return dst


def propagate(feats, ref_seg_map, out_hw, *, precede_frames=20, topk=10, temperature=0.07, neighbor_range=None,
              with_first=True, with_first_neighbor=True, with_norm=True, return_logits=False):
    """vanilla_tracker.py:94-181 given the UN-normalised feature maps feats [T,h,w,C] (NHWC fp32) and the first
    frame's uint8 labels [H,W] (-> uint8 [T,H_out,W_out]) or a one-hot / soft map [CO,H,W] float (-> float
    [T,CO,H_out,W_out], no min-max / argmax: vanilla_tracker.py:167)."""
    T, h, w, C = feats.shape
    flat = f32(feats).reshape(T * h * w, C)
    bank = (l2norm_rows(flat) if with_norm else flat).reshape(T, h * w, C)
    onehot = ref_seg_map.ndim == 3
    if onehot:
        CO = ref_seg_map.shape[0]
        sbank = np.zeros((T, h * w, CO), np.float32)
        sbank[0] = bilinear_resize(ref_seg_map, h, w, False, True).reshape(h * w, CO)
        preds = np.empty((T, CO) + tuple(out_hw), np.float32)
        preds[0] = bilinear_resize(ref_seg_map, out_hw[0], out_hw[1], False, False)
    else:
        small = pil_nearest_resize(ref_seg_map, h, w)
        CO = int(small.max()) + 1
        sbank = np.zeros((T, h * w, CO), np.float32)
        sbank[0] = np.eye(CO, dtype=np.float32)[small.reshape(-1)]
        preds = np.empty((T,) + tuple(out_hw), np.uint8)
        preds[0] = torch_nearest_resize(ref_seg_map, *out_hw)
    radius = int(neighbor_range) // 2 if neighbor_range is not None else 0
    for f in range(1, T):
        ks = max(0, f - precede_frames)
        slots = list(range(ks, f))
        if with_first:
            slots = [0] + slots
        sbank[f] = labelprop(bank, sbank, f, slots, h, w, radius, topk, temperature, non_mask_len=0 if with_first_neighbor else 1)
        if onehot:
            preds[f] = bilinear_resize(sbank[f].reshape(h, w, CO), out_hw[0], out_hw[1], True, False)
        else:
            preds[f] = seg_postprocess(sbank[f], h, w, out_hw[0], out_hw[1])
    return (preds, sbank, bank) if return_logits else preds


def forward_test(sd, depth, imgs, ref_seg_map, original_shape, test_cfg, prefix='backbone.'):
    """VanillaTracker.forward_test (vanilla_tracker.py:80-206) for a state_dict with the reference's names.
    imgs [1,1,3,T,H,W]; returns uint8 [T,H,W] (or [num_feats,T,H,W] with test_cfg.all_blocks)."""
    imgs = _np(imgs)
    imgs = imgs.reshape((-1,) + imgs.shape[2:])                    # [1,3,T,H,W]
    frames = np.ascontiguousarray(np.transpose(imgs[0], (1, 0, 2, 3)))   # [T,3,H,W]
    tc = test_cfg
    step = int(tc.get('batch_step', 10))
    all_blocks = bool(tc.get('all_blocks', False))
    chunks = [resnet_eval(sd, depth, frames[i:i + step], strides=tuple(tc.get('strides', (1, 2, 1, 1))),
                          out_indices=tuple(tc.get('out_indices', (2,))), prefix=prefix, all_blocks=all_blocks)
              for i in range(0, frames.shape[0], step)]
    if all_blocks:
        feats = [np.concatenate([c[i] for c in chunks]) for i in range(len(chunks[0]))]
    else:
        si = tuple(tc.get('out_indices', (2,)))[0]
        feats = [np.concatenate([c[si] for c in chunks])]
    ref = _np(ref_seg_map)
    res = []
    for f in feats:
        res.append(propagate(f, ref, tuple(original_shape[:2]), precede_frames=int(tc['precede_frames']), topk=int(tc['topk']),
                             temperature=float(tc['temperature']), neighbor_range=tc.get('neighbor_range'),
                             with_first=tc.get('with_first', True), with_first_neighbor=tc.get('with_first_neighbor', True),
                             with_norm=tc.get('with_norm', True)))
        ref = res[-1][0]        # the reference keeps the resized first-frame map for the next feature level (:101-111)
    return np.stack(res, 0) if all_blocks else res[0]
