#!/usr/bin/env python3
"""Two-pass label propagation on the GPU at DAVIS size: list fill (candidates pass 1 listed per query and key-frame split), overflow
flag, kernel times.  usage: python tools/lp2_stats.py [r50|r18] [cap]"""
import ctypes
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import vfs_amd                                       # noqa: E402
from vfs_amd.engine import shared_engine             # noqa: E402
from vfs_amd.labelprop import extract_features       # noqa: E402
from vfs_amd.synthetic import synthetic_weights_     # noqa: E402

model_name = sys.argv[1] if len(sys.argv) > 1 else 'r50'
depth = 18 if model_name == 'r18' else 50
dev = torch.device('cuda:0')
cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_r{depth}.py'))
tc = vfs_amd.ConfigDict(cfg.test_cfg)
bb = dict(cfg.model['backbone'])
bb['out_indices'], bb['strides'] = tc['out_indices'], tc['strides']
model = vfs_amd.build_model(dict(type='VanillaTracker', backbone=bb), train_cfg=None, test_cfg=tc)
synthetic_weights_(model, seed=5)
model.to(dev).eval()
eng = shared_engine()
lib = eng.lib
if len(sys.argv) > 2 and int(sys.argv[2]) > 0:
    lib.set_option(b'lp2_cap', int(sys.argv[2]))
T, H, W = 22, 480, 854
g = torch.Generator(device=dev).manual_seed(1234)
base = torch.randn(1, 3, 1, H, W, device=dev, generator=g)
imgs = base + 0.15 * torch.randn(1, 3, T, H, W, device=dev, generator=g)       # bench.py's clip
hls = []
bank, h, w, C = extract_features(model, eng, imgs, 10, precision='fp32', split_banks=hls)
hl = hls[0]
CO = 4
sbank = torch.rand(T, h * w, CO, device=dev)
radius = int(tc['neighbor_range']) // 2
n = torch.zeros(1, dtype=torch.int64)
lib.labelprop_f32_2pass_workspace_bytes(h, w, n)
dense = torch.zeros(1, dtype=torch.int64)
lib.labelprop_workspace_bytes(h, w, dense)
ws = torch.zeros((int(n.item()) + 3) // 4, device=dev)
f = T - 1
slots = [0] + list(range(f - 20, f))
ks = (ctypes.c_int * len(slots))(*slots)
out2 = torch.empty(h * w, CO, device=dev)
out1 = torch.empty(h * w, CO, device=dev)
s = eng.stream(dev)


def two():
    lib.labelprop_f32_2pass(bank, hl, sbank, out2, ws, ws.numel() * 4, f, ks, len(slots), h, w, C, CO, radius, 0, 10, 0.07, 1, s)


def one():
    lib.labelprop_f32(bank, sbank, out1, ws, ws.numel() * 4, f, ks, len(slots), h, w, C, CO, radius, 0, 10, 0.07, s)


for fn, name in ((two, 'two-pass'), (one, 'dense')):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per frame ({len(slots)} key frames, C = {C}, radius {radius})')
two()
torch.cuda.synchronize()
print('bit-equal to the dense kernel:', bool(torch.equal(out1.view(torch.int32), out2.view(torch.int32))))
HW = h * w
lists_bytes = 24 * HW * 192 * 8
counts = ws.view(torch.int32)[(int(dense.item()) + lists_bytes) // 4:(int(dense.item()) + lists_bytes) // 4 + 24 * HW].reshape(24, HW)
flag = int(ws.view(torch.int32)[(int(n.item()) - 16) // 4])
per_q = counts.sum(0).float()
print('overflow flag', flag, '| listed per query: mean %.0f max %d | per (split, query): max %d | splits used %d'
      % (per_q.mean(), int(per_q.max()), int(counts.max()), int((counts.sum(1) > 0).sum())))

if os.environ.get('LP2_PROF'):      # a build with phase timers in pass 1 (see MEASUREMENTS.md round 4): clk per wave of one mid workgroup
    row = ws.view(torch.int32)[(int(dense.item()) + lists_bytes) // 4 + 23 * HW:(int(dense.item()) + lists_bytes) // 4 + 23 * HW + 64].cpu().view(torch.int64).reshape(4, 8)
    names = ['kernel', 'stage loops', ' of which DMA waits', 'park + barrier 1', 'reduce + threshold', 'barrier 2', 'wave-0 top-10 / tail', 'blocks']
    for w in range(4):
        print('wave', w, ', '.join(f'{n} {int(v)}' for n, v in zip(names, row[w].tolist())))
if len(sys.argv) > 3:      # only the step with that many key frames, 20 times (for `rocprofv3 --kernel-trace --stats -- python tools/lp2_stats.py r50 0 <nkeys>`)
    nk = int(sys.argv[3])
    fq = nk - 1
    sl = [0] + list(range(max(0, fq - 20), fq))
    kk = (ctypes.c_int * len(sl))(*sl)
    for _ in range(20):
        lib.labelprop_f32_2pass(bank, hl, sbank, out2, ws, ws.numel() * 4, fq, kk, len(sl), h, w, C, CO, radius, 0, 10, 0.07, 1, s)
    torch.cuda.synchronize()
    sys.exit(0)
# every propagation step of a clip (first + up to 20 preceding key frames, as forward_test builds them): time and fallback flag
print('key frames : two-pass ms, dense-fallback flag')
for fq in range(1, T):
    sl = [0] + list(range(max(0, fq - 20), fq))
    kk = (ctypes.c_int * len(sl))(*sl)
    def step():
        lib.labelprop_f32_2pass(bank, hl, sbank, out2, ws, ws.numel() * 4, fq, kk, len(sl), h, w, C, CO, radius, 0, 10, 0.07, 1, s)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    fl = int(ws.view(torch.int32)[(int(n.item()) - 16) // 4])
    cnts = ws.view(torch.int32)[(int(dense.item()) + lists_bytes) // 4:(int(dense.item()) + lists_bytes) // 4 + 24 * HW].reshape(24, HW)
    pq = cnts.sum(0).float()
    print(f'{len(sl):3d} : {dt:6.3f} ms  flag {fl}  listed per query mean {float(pq.mean()):7.1f} max {int(pq.max())}  splits used {int((cnts.sum(1) > 0).sum())}')
