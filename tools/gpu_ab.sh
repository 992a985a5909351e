#!/bin/bash
# Whole-step A/B on one box: each argument is one arm, a string of environment assignments ("-" = defaults), e.g.
#   tools/gpu_ab.sh - "VFS_OPTS=igemm_ring_tiles=4096" "VFS_WGRAD_TB=192 VFS_HIP_LIB=$PWD/tools/_build/libvfs_b.so"
# (VFS_OPTS = library knobs, VFS_HIP_LIB = another build of the same ABI, any VFS_* switch of DESIGN.md section 10).
# Two interleaved rounds over both models; output also in gpurun_out/${TAG}_ab.txt.  MODELS="r50" restricts the models.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-ab}
B="--steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-roofline --no-davis"
{
for i in 1 2; do for E in "$@"; do for M in ${MODELS:-r50 r18}; do
  if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
  echo -n "$M [$E]: "; env $EE timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done; done; done
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
