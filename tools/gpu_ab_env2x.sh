#!/bin/bash
# A/B of environment settings on the whole step: tools/gpu_ab_env2x.sh "<VAR=val ...>" "<VAR=val ...>" ...  ("-" = defaults)
cd "$GRAFT_REPO_ROOT"
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do for E in "$@"; do for M in r50 r18; do
  if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
  echo -n "$M [$E]: "; env $EE timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done; done; done
