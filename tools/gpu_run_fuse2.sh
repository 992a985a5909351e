#!/bin/bash
cd $GRAFT_REPO_ROOT
for M in r18 r50; do for i in 1 2; do for V in 0 24 48 100 100000; do
  echo -n "$M VFS_BNACT_FUSE_MB=$V: "; VFS_BNACT_FUSE_MB=$V timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
done; done; done
