#!/bin/bash
# A/B of library options on the whole step: tools/gpu_ab_opts.sh "<opts1>" "<opts2>" ...   (each a VFS_OPTS string, "-" = default)
cd "$GRAFT_REPO_ROOT"
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for i in 1 2; do for O in "$@"; do for M in r50 r18; do
  if [ "$O" = "-" ]; then OO=""; else OO="$O"; fi
  echo -n "$M [$O]: "; VFS_OPTS="$OO" timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done; done; done
