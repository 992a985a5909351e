#!/bin/bash
# kernel statistics of the step WITHOUT collectives and in a 1-rank group with the fused window exchange: where does the N > 1 code path spend its extra time?
TAG=${1:-r03_d}
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
for MODE in none p2p; do
  if [ $MODE = p2p ]; then export VFS_FORCE_COLLECTIVES=1 VFS_SYNCBN_P2P=force; fi
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$MODE -o r50 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-davis > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$MODE.log 2>&1
  cd $GRAFT_REPO_ROOT
  cp $(find gpurun_out/${TAG}_prof_$MODE -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_r50_${MODE}_kernel_stats.csv
  rm -rf gpurun_out/${TAG}_prof_$MODE
  grep "timed steps" gpurun_out/${TAG}_prof_$MODE.log
done
