#!/bin/bash
cd "$GRAFT_REPO_ROOT"
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for M in r18 r50 r18 r50; do
  echo -n "$M: "; timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done
echo -n "r18 100 steps: "; timeout 300 python bench.py --model r18 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
echo -n "r50 100 steps: "; timeout 300 python bench.py --model r50 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
