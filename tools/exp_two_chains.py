#!/usr/bin/env python3
"""Experiment (measurement only): what would TWO INDEPENDENT launch chains on two streams buy?

The two augmented views of a SimSiam step are independent up to the loss (separate BatchNorm batches, stop-gradient
targets), but the engine issues them as ONE chain of launches over both views.  ~45 % of that chain's launches are short
(<= 15 us: statistics finalisers, the BatchNorm passes and 1x1 convolutions of layers 3-4, the head) and sit at the ~5 us
launch floor.  Before restructuring the engine into per-view chains, this script measures the upper bound cheaply: two
complete models with half the batch each (their own engine, buffers, streams), stepped alternately so that the GPU sees two
independent chains - against one model with the full batch.  Same total frames per iteration.

    python tools/exp_two_chains.py [--model r50] [--batch 32] [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('VFS_GC_FREEZE', '1')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='r50')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--steps', type=int, default=20)
    args = ap.parse_args()
    import vfs_amd
    from vfs_amd import engine
    dev = torch.device('cuda:0')
    depth = 18 if args.model == 'r18' else 50
    cfg = vfs_amd.Config.fromfile(os.path.join(REPO, 'configs', f'vfs_r{depth}.py'))
    T = int(cfg.clip_len)

    class Chain:
        def __init__(self, B, seed):
            self.eng = engine.Engine()
            engine.set_shared_engine(self.eng)
            torch.manual_seed(0)
            self.model = vfs_amd.build_model(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).to(dev).train()
            self.model.flatten_parameters()
            self.opt = vfs_amd.build_optimizer(self.model, cfg.optimizer)
            self.stream = torch.cuda.Stream(dev)
            g = torch.Generator(device=dev).manual_seed(seed)
            self.batch = dict(imgs=torch.randn(B, 2, 3, T, args.size, args.size, device=dev, generator=g), label=torch.zeros(B, 1, device=dev))
            self.out = None

        def fwd(self):
            engine.set_shared_engine(self.eng)
            with torch.cuda.stream(self.stream):
                self.out = self.model.train_step(self.batch, self.opt)
                self.opt.zero_grad()

        def bwd(self):
            engine.set_shared_engine(self.eng)
            with torch.cuda.stream(self.stream):
                self.out['loss'].backward()
                self.opt.step()

    def run(chains, steps):
        for c in chains:                 # eager pass + recording pass
            for _ in range(3):
                c.fwd()
                c.bwd()
        torch.cuda.synchronize()
        for _ in range(3):
            for c in chains:
                c.fwd()
            for c in chains:
                c.bwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for c in chains:
                c.fwd()
            for c in chains:
                c.bwd()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    one = run([Chain(args.batch, 1)], args.steps)
    half_alone = run([Chain(args.batch // 2, 2)], args.steps)
    two = run([Chain(args.batch // 2, 3), Chain(args.batch // 2, 4)], args.steps)
    print(f'{args.model} {args.size}^2: one chain, B={args.batch}: {one:.2f} ms/step | one chain, B={args.batch // 2}: {half_alone:.2f} ms | '
          f'TWO chains of B={args.batch // 2} on two streams: {two:.2f} ms per pair of steps (same frames as the first figure)')


if __name__ == '__main__':
    main()
