#!/bin/bash
# round 3, third GPU call: whole-step A/B of the XCD-aware block orders (weight gradients, halo kernels), the cost of the
# collective path in a 1-rank group (RCCL vs the P2P window exchange), the failing golden test again.  usage: <tag>
TAG=${1:-r03_c}
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-davis"
{
for i in 1 2; do for O in "-" "wgrad_xcd=0" "halo_xcd=0" "wgrad_xcd=0,halo_xcd=0"; do for M in r50 r18; do
  if [ "$O" = "-" ]; then OO=""; else OO="$O"; fi
  echo -n "$M [$O]: "; VFS_OPTS="$OO" timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done; done; done
for M in r50 r18; do
  echo -n "$M 1-rank group, RCCL SyncBN all-reduces: "; VFS_FORCE_COLLECTIVES=1 VFS_SYNCBN_P2P=0 timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps|rror" | sed 's/.*timed steps: //'
  echo -n "$M 1-rank group, P2P window exchange:     "; VFS_FORCE_COLLECTIVES=1 VFS_SYNCBN_P2P=force timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps|rror|P2P" | sed 's/.*timed steps: //'
done
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
timeout 300 python -m pytest tests/test_cfg1_golden.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
