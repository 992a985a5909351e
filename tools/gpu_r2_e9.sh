#!/bin/bash
# does the weight repack sit on the critical path?  new build (vectorised pointwise tiles, 60 us) / base build (164 us) / no repack at all
cd "$GRAFT_REPO_ROOT"
B="--model r50 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  echo -n "new:  "; timeout 300 python bench.py $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
  echo -n "base: "; VFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/_bin/libvfs_base.so timeout 300 python bench.py $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
  echo -n "skip: "; VFS_DEBUG_SKIP=pack_weights timeout 300 python bench.py $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done
