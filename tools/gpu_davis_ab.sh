#!/bin/bash
# DAVIS label propagation A/B: each argument is one arm (environment assignments, "-" = defaults); fp32 path of both models
# (CASES="r50 fp32;r50 bf16;r18 fp32" to choose).  Output also in gpurun_out/${TAG}_davis_ab.txt
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-davis}
IFS=';' read -ra CS <<< "${CASES:-r50 fp32;r18 fp32}"
{
for i in 1 2; do for E in "$@"; do for C in "${CS[@]}"; do
  M=${C% *}; P=${C#* }
  if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
  echo -n "$M $P [$E]: "; env $EE timeout 300 python bench.py --workload davis --model $M --precision $P --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "ms/frame" | sed 's/.*frames: //'
done; done; done
} 2>&1 | tee gpurun_out/${TAG}_davis_ab.txt
