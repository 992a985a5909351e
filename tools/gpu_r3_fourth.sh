#!/bin/bash
# round 3, fourth GPU call: the SyncBN reductions with the window exchange as their tail - two processes on one GPU (bit-equal to the
# collective path), and what the N > 1 code path costs on one GPU (1-rank group): RCCL all-reduces vs the fused exchange.
TAG=${1:-r03_d}
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_p2p.py tests/test_cfg1_golden.py -m gpu -q -p no:cacheprovider -s 2>&1 | tail -6 | cut -c1-300
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-davis"
{
for i in 1 2; do for M in r50 r18; do
  echo -n "$M no collectives:                                  "; timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
  echo -n "$M 1-rank group, RCCL SyncBN all-reduces:           "; VFS_FORCE_COLLECTIVES=1 VFS_SYNCBN_P2P=0 timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps|rror" | sed 's/.*timed steps: //'
  echo -n "$M 1-rank group, window exchange fused in reducers: "; VFS_FORCE_COLLECTIVES=1 VFS_SYNCBN_P2P=force timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps|rror|P2P" | sed 's/.*timed steps: //'
done; done
} 2>&1 | tee gpurun_out/${TAG}_collectives.txt
