#!/bin/bash
# everything the committed profiles/ are made of: bench JSON lines (both models), rocprofv3 kernel stats,
# HBM traffic from the PMC passes.  usage: tools/gpu_profiles.sh <tag>
TAG=${1:-r02_x}
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for M in ${2:-r18 r50}; do
  ./tools/gpu_pmc.sh $M > gpurun_out/${TAG}_pmc_$M.txt 2>&1
  python tools/make_traffic_json.py $M $TAG >> gpurun_out/${TAG}_pmc_$M.txt 2>&1
  cp profiles/${TAG}_traffic_$M.json profiles/r03_traffic_$M.json
  cp profiles/${TAG}_traffic_$M.json gpurun_out/
  timeout 600 python bench.py --model $M --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_$M.json 2> gpurun_out/${TAG}_bench_$M.log
  tail -3 gpurun_out/${TAG}_bench_$M.log; cat gpurun_out/${TAG}_bench_$M.json
  ./tools/gpu_prof.sh $M ${TAG}_prof_$M > gpurun_out/${TAG}_prof_$M.txt 2>&1
  cp $(find gpurun_out/${TAG}_prof_$M -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_${M}_bench_kernel_stats.csv
  head -12 gpurun_out/${TAG}_prof_$M.txt
done
