"""Worker of tests/test_dist_gloo.py: one data-parallel rank on the CPU (gloo) running the
fused train step through the fiber emulator.  Usage: dist_worker.py OUT.npz (env RANK/WORLD_SIZE/...)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def build(world_batch):
    import vfs_amd
    from oracle import vfs_oracle as O
    from tests.test_emu_train_step import SHALLOW_HEAD, SHALLOW_MINE, _filled
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', 'vfs_r18.py'))
    mcfg = dict(cfg.model)
    mcfg['backbone'] = dict(mcfg['backbone'], **SHALLOW_MINE)
    mcfg['img_head'] = dict(mcfg['img_head'], **SHALLOW_HEAD)
    model = vfs_amd.build_model(mcfg, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    model.load_state_dict(_filled(18, shallow=True).state_dict())
    model.train()
    imgs = O.fill_tensor([world_batch, 2, 3, 2, 32, 32], seed=11, scale=2.0)
    return model, imgs, cfg


def run(rank, world, out_path):
    from tests.emu_util import emu_lib
    from vfs_amd import engine
    torch.set_num_threads(2)
    eng = engine.Engine(lib=emu_lib())
    engine.set_shared_engine(eng)
    model, imgs, cfg = build(int(os.environ.get('VFS_TEST_BATCH', '8')))
    per = imgs.shape[0] // world
    local = imgs[rank * per:(rank + 1) * per]
    import vfs_amd
    opt = vfs_amd.build_optimizer(model, cfg.optimizer)
    # VFS_TEST_STEPS > 1: the later steps run from the recorded command tapes (or eagerly with VFS_TAPE=0)
    for step in range(int(os.environ.get('VFS_TEST_STEPS', '1'))):
        batch = local if step == 0 else (local * (1.0 + 0.25 * step)).contiguous()
        out = model.train_step(dict(imgs=batch, label=torch.zeros(per, 1)), None)
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
    res = {'log/' + k: np.float64(v) for k, v in out['log_vars'].items()}
    for n, p in model.named_parameters():
        res['grad/' + n] = p.grad.detach().numpy().copy()
        res['param/' + n] = p.detach().numpy().copy()
    for n, b in model.named_buffers():
        if 'running' in n:
            res['buf/' + n] = b.numpy().copy()
    np.savez(out_path, **res)


if __name__ == '__main__':
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    run(rank, world, sys.argv[1])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
