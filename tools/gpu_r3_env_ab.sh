#!/bin/bash
# whole-step A/B of environment switches: tools/gpu_r3_env_ab.sh "<VAR=val ...>" "<...>"   ("-" = defaults)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-r03_env}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-davis"
{
for i in 1 2; do for E in "$@"; do for M in r50 r18; do
  if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
  echo -n "$M [$E]: "; env $EE timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done; done; done
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
