#!/bin/bash
# A/B of the weight-gradient split plans (fewer splits = less fp32 partial traffic, fewer workgroups) on the whole step
cd $GRAFT_REPO_ROOT
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for M in r50 r18; do
for cfg in "512 1024" "256 1024" "128 1024" "512 512" "512 256" "256 512" "256 256" "128 256" "128 128" "64 128"; do
  set -- $cfg
  echo -n "$M TB=$1 TBG=$2: "; VFS_WGRAD_TB=$1 VFS_WGRAD_TBG=$2 timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done; done
