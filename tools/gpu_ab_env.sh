#!/bin/bash
# A/B of an environment switch on one box: tools/gpu_ab_env.sh VAR valueA valueB [model]
cd $GRAFT_REPO_ROOT
M=${4:-r18}
for i in 1 2; do for V in $2 $3; do
  echo -n "$1=$V: "; env $1=$V timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
done; done
