#!/bin/bash
# DAVIS label propagation: A/B of library options (each a VFS_OPTS string, "-" = default), fp32 and bf16 evaluation paths
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-r03_f}
{
for i in 1 2; do for O in "$@"; do for C in "r50 fp32" "r50 bf16" "r18 fp32"; do
  M=${C% *}; P=${C#* }
  if [ "$O" = "-" ]; then OO=""; else OO="$O"; fi
  echo -n "$M $P [$O]: "; VFS_OPTS="$OO" timeout 300 python bench.py --workload davis --model $M --precision $P --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "ms/frame" | sed 's/.*frames: //'
done; done; done
} 2>&1 | tee gpurun_out/${TAG}_davis_ab.txt
