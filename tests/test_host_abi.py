"""CPU checks of the boundary: the C-ABI library loads and exports every symbol the header
declares, configs (ours and -- when present -- the reference's own files) build through the
registry with the reference's state_dict names, error behaviour matches the reference's tests."""
import os

import pytest
import torch

import vfs_amd
from vfs_amd import _lib, build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = '/root/reference/configs'


def test_library_exports_every_declared_symbol():
    path = build.build_hip()
    protos = _lib.parse_header()
    assert len(protos) >= 25 and 'vfs_conv_fwd' in protos and 'vfs_last_error' in protos
    lib = _lib.VfsLib(path)                     # resolves every prototype or raises
    assert lib.dll.vfs_abi_version() == 2
    import subprocess
    syms = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True).stdout
    for name in protos:
        assert f' T {name}' in syms, name


def test_labelprop_workspace_query_is_host_only():
    """vfs_labelprop_workspace_bytes: the size contract of the label propagation workspace (no GPU involved), and its argument check"""
    lib = _lib.VfsLib(build.build_hip())
    n = torch.zeros(1, dtype=torch.int64)
    lib.labelprop_workspace_bytes(60, 107, n)
    assert n.item() == 96 * 60 * 107 * 10 * 8
    with pytest.raises(_lib.VfsError, match='labelprop_workspace_bytes'):
        lib.labelprop_workspace_bytes(0, 107, n)
    with pytest.raises(_lib.VfsError):
        lib.labelprop_workspace_bytes(60, 107, None)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.VfsError):
        _lib.VfsLib(str(tmp_path / 'libvfs_hip.so'))


@pytest.mark.parametrize('depth', [18, 50])
def test_own_configs_build_with_reference_state_dict_names(depth, golden_dir):
    import numpy as np
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_r{depth}.py'))
    model = vfs_amd.build_model(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    g = np.load(os.path.join(golden_dir, f'r{depth}_train.npz'))
    assert list(model.state_dict().keys()) == [str(k) for k in g['keys']]   # captured from the reference
    # init statistics of the reference model (kaiming fan_out convs, BN 1/0, zero-init residual)
    sd = model.state_dict()
    for k, v in sd.items():
        key = 'init/' + k
        if key in g.files and v.numel() > 1:
            m, s = float(v.float().mean()), float(v.float().std())
            rm, rs = g[key]
            if k.endswith('bn.weight') or k.endswith('bn.bias') or 'running' in k:
                assert abs(m - rm) < 1e-6 and abs(s - rs) < 1e-6, k
            elif k.endswith('conv.weight'):
                assert abs(s - rs) < 0.1 * rs + 1e-4, k      # same distribution, different RNG draw
    assert model.intra_video == (depth == 18)


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference checkout not present')
@pytest.mark.parametrize('name', ['r18_nc_sgd_cos_100e_r2_1xNx8_k400.py', 'r50_nc_sgd_cos_100e_r5_1xNx2_k400.py',
                                  'r18_sgd_cos_100e_r2_1xNx8_k400.py', 'r50_sgd_cos_100e_r5_1xNx2_k400.py'])
def test_reference_configs_load_unchanged(name):
    cfg = vfs_amd.Config.fromfile(os.path.join(REF_CFG, name))
    model = vfs_amd.build_model(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert type(model).__name__ == 'SimSiamBaseTracker'
    own = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', 'vfs_r18.py' if name.startswith('r18') else 'vfs_r50.py'))
    strip = lambda d: {k: (strip(v) if isinstance(v, dict) else v) for k, v in d.items()}
    assert strip(cfg.model) == strip(own.model)
    assert dict(cfg.train_cfg) == dict(own.train_cfg) and dict(cfg.test_cfg) == dict(own.test_cfg)
    assert dict(cfg.optimizer) == dict(own.optimizer)
    # tools/test.py:129-133 rebuilds the model as a VanillaTracker with the test-time strides
    bb = dict(cfg.model['backbone'])
    bb['out_indices'], bb['strides'] = cfg.test_cfg.out_indices, cfg.test_cfg.strides
    vt = vfs_amd.build_model(dict(type='VanillaTracker', backbone=bb), train_cfg=None, test_cfg=cfg.test_cfg)
    assert vt.stride == 8


def test_resnet_constructor_errors_like_the_reference():
    """tests/test_models/test_backbone.py:24-49 of the reference."""
    with pytest.raises(KeyError):
        vfs_amd.ResNet(20)
    with pytest.raises(AssertionError):
        vfs_amd.ResNet(50, num_stages=0)
    with pytest.raises(AssertionError):
        vfs_amd.ResNet(50, num_stages=5)
    with pytest.raises(AssertionError):
        vfs_amd.ResNet(50, strides=(1,), dilations=(1, 1), num_stages=3)
    with pytest.raises(TypeError):
        net = vfs_amd.ResNet(50, pretrained=0)
        net.init_weights()
    net = vfs_amd.ResNet(18, norm_eval=True)
    net.init_weights()
    net.train()
    assert all(not m.training for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))
    net = vfs_amd.ResNet(50, frozen_stages=1)
    net.train()
    assert not net.conv1.bn.training and all(not p.requires_grad for p in net.layer1.parameters())
    with pytest.raises(KeyError):
        vfs_amd.build_model(dict(type='NoSuchTracker'))


def test_forward_train_asserts_like_the_reference():
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', 'vfs_r18.py'))
    model = vfs_amd.build_model(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    with pytest.raises(AssertionError):
        model.forward_train(torch.zeros(2, 3, 3, 1, 32, 32))      # imgs.size(1) must be 2
    with pytest.raises(AssertionError):
        model.forward_train(torch.zeros(2, 2, 3, 32, 32))         # must be 6-D


def test_torchvision_checkpoint_key_mapping(tmp_path):
    """resnet.py:488-523 + tools/convert_weights/convert_to_pretrained.py naming."""
    net = vfs_amd.ResNet(18)
    tv = {}
    for name, m in net.conv_modules():
        cname, bname = (name + '.0', name + '.1') if 'downsample' in name else (name, name.replace('conv', 'bn'))
        tv[cname + '.weight'] = torch.randn_like(m.conv.weight)
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            tv[f'{bname}.{k}'] = torch.randn(m.bn.num_features)
    f = tmp_path / 'tv.pth'
    torch.save(dict(state_dict=tv), f)
    net2 = vfs_amd.ResNet(18, pretrained=str(f))
    net2.init_weights()
    assert torch.equal(net2.layer2[0].downsample.conv.weight, tv['layer2.0.downsample.0.weight'])
    assert torch.equal(net2.layer3[1].conv2.bn.running_var, tv['layer3.1.bn2.running_var'])
    assert torch.equal(net2.conv1.conv.weight, tv['conv1.weight'])


def test_command_tape_records_and_checks_return_codes():
    """_lib.Tape / TapeLib: calls are recorded with converted arguments, replayed in order, and a failing entry point
    raises both while recording and on replay (host logic only: vfs_set_option launches nothing)"""
    from vfs_amd._lib import Tape, TapeLib, VfsError, get_lib
    lib = get_lib()
    tape = Tape(lib)
    rec = TapeLib(lib, tape)
    seen = []
    rec.set_option(b'halo', 1)
    tape.ops.append((None, seen.append, ('py',)))          # a Python-side action between two C calls
    rec.set_option(b'bn_ticket', 1)
    assert [op[0] for op in tape.ops] == ['set_option', None, 'set_option'] and tape.ops[0][2] == (b'halo', 1)
    tape.replay()
    assert seen == ['py']
    with pytest.raises(VfsError):
        rec.set_option(b'no_such_option', 1)
    with pytest.raises(VfsError):
        tape.replay()                                       # the failing call was recorded too
    with pytest.raises(AttributeError):
        rec.no_such_entry_point


def test_no_compiler_generated_m0_use_in_lds_dma_kernels(tmp_path):
    """advisor r05: vfs_dma16_async_at (csrc/vfs_common.h) writes m0 from inline asm without saving it and declares the clobber,
    which clang warns it cannot honour for a reserved register.  That is safe exactly as long as the COMPILER keeps nothing of its own
    in m0 in those kernels (s_movrel / v_movrel / s_set_gpr_idx indexing, readlane-by-m0, LDS-direct, builtin LDS-DMA).  Build-time
    check: every source that issues LDS-DMA pieces is compiled to gfx950 assembly and each line that mentions m0 must be one of
    ours - `s_mov_b32 m0, <lds address>` in front of a `buffer_load_dword* ... lds`, or the save / restore pair of
    vfs_dma16_async."""
    import re
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    srcs = [s for s in build._sources() if re.search(r'vfs_dma16_async', open(s).read())]
    assert len(srcs) >= 4

    def asm(src):
        out = tmp_path / (os.path.basename(src) + '.s')
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-w', '-o', str(out), src])
        return src, out.read_text().splitlines()

    with ThreadPoolExecutor(4) as ex:
        for src, lines in ex.map(asm, srcs):
            code = [ln.split(';')[0].strip() for ln in lines]
            code = [c for c in code if c and not c.startswith('.') and not c.endswith(':')]      # instructions only
            mine = 0
            for i, c in enumerate(code):
                if not re.search(r'\bm0\b', c):
                    continue
                nxt, prv = code[i + 1:i + 3], code[max(0, i - 1):i]
                setup = re.match(r's_mov_b32 m0, \S+$', c) and len(nxt) == 2 and nxt[0].startswith('s_nop') and \
                    re.match(r'buffer_load_dword\S* .* lds$', nxt[1])                                          # ours: m0 <- LDS address, s_nop, the DMA piece
                save = re.match(r's_mov_b32 \S+, m0$', c) and nxt and re.match(r's_mov_b32 m0, \S+$', nxt[0])      # vfs_dma16_async: save ...
                restore = re.match(r's_mov_b32 m0, \S+$', c) and prv and re.match(r'buffer_load_dword\S* .* lds$', prv[0])      # ... and restore
                step = re.match(r's_add_u32 m0, m0, 0x[0-9a-f]+$', c) and len(nxt) == 2 and nxt[0].startswith('s_nop') and \
                    re.match(r'buffer_load_dword\S* .* lds$', nxt[1])      # labelprop2.hip: eight pieces of one stage, m0 stepped in place
                assert setup or save or restore or step, f'{os.path.basename(src)}: compiler-generated m0 use: {code[max(0, i - 2):i + 3]}'
                mine += 1
            assert mine > 0, f'{os.path.basename(src)}: no LDS-DMA m0 set-up found (the check looks at the wrong thing)'
