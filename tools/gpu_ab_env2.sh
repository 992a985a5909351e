#!/bin/bash
# A/B of host-side switches on the whole step: each argument is "VAR=value[,VAR=value...]" ("-" = defaults)
cd "$GRAFT_REPO_ROOT"
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for i in 1 2; do for E in "$@"; do
  if [ "$E" = "-" ]; then EE=""; else EE=$(echo "$E" | tr ',' ' '); fi
  echo -n "r50 [$E]: "; env $EE timeout 300 python bench.py --model r50 $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done; done
