#!/usr/bin/env python3
"""Which host-side ops issue the small device copies seen in the kernel trace?"""
import os, sys, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['VFS_GRAPHS'] = '0'
import vfs_amd
from vfs_amd.optim import build_optimizer

cfg = vfs_amd.Config.fromfile(os.path.join(os.path.dirname(__file__), '..', 'configs', 'vfs_r18.py'))
model = vfs_amd.build_model(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).cuda().train()
opt = build_optimizer(model, cfg.optimizer)
imgs = torch.randn(8, 2, 3, 4, 64, 64, device='cuda')
def step():
    out = model.train_step(dict(imgs=imgs), opt)
    opt.zero_grad(); out['loss'].backward(); opt.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
counts = collections.Counter()
orig = torch.Tensor.copy_
def traced_copy(self, src, *a, **k):
    if self.is_cuda:
        st = traceback.extract_stack(limit=6)[:-1]
        counts[' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st[-3:]))] += 1
    return orig(self, src, *a, **k)
torch.Tensor.copy_ = traced_copy
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
torch.Tensor.copy_ = orig
print('python-level Tensor.copy_ call sites:', dict(counts))
rows = [(e.key, e.count) for e in prof.key_averages() if any(s in e.key.lower() for s in ('copy', 'memcpy', 'clone', 'contiguous', 'to', 'fill', 'zero', 'add', 'mul', 'cat'))]
for k, c in sorted(rows, key=lambda kv: -kv[1])[:30]:
    print(f'{c:5d}  {k}')
