#!/bin/bash
TAG=${1:-r01_x}
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
for M in r18 r50; do
  timeout 600 python bench.py --model $M --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_$M.json 2> gpurun_out/${TAG}_bench_$M.log
  tail -2 gpurun_out/${TAG}_bench_$M.log
done
