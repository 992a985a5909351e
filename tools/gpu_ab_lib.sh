#!/bin/bash
# A/B of two builds of the library on one box: vfs_amd/csrc/libvfs_hip.so (new) vs tools/_bin/libvfs_base.so
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for M in ${1:-r50 r18}; do
  echo -n "new  $M: "; timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
  echo -n "base $M: "; VFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/_bin/libvfs_base.so timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
done; done
