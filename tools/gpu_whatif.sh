#!/bin/bash
# what does each kernel family cost on the critical path?  step time with its launches dropped (VFS_DEBUG_SKIP)
cd "$GRAFT_REPO_ROOT"
M=${1:-r50}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-davis"
for S in "" "wgrad_reduce" "conv_wgrad,conv3x3_wgrad_halo,stem_wgrad,wgrad_reduce" "conv_wgrad,conv3x3_wgrad_halo,stem_wgrad" "bn_act" "bn_bwd_apply" "bn_bwd_reduce" "bn_stats" "conv_igemm" "conv3x3_halo" "pack_weights" "sgd" "bn_act,bn_bwd_apply,bn_bwd_reduce,bn_stats,bn_relu_maxpool" "conv_igemm,conv3x3_halo,stem_fwd"; do
  echo -n "$M skip [$S]: "; VFS_DEBUG_SKIP="$S" timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done
