#!/bin/bash
# what-if: the step without the mask-operand reads of BatchNorm backward (upper bound for a bit-packed mask)
cd "$GRAFT_REPO_ROOT"
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do for M in r50 r18; do
  echo -n "$M base:   "; timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
  echo -n "$M nomask: "; VFS_DEBUG_NOMASK=1 timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done; done
