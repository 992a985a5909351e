#!/bin/bash
# The committed evidence of a round, one call: tools/gpu_evidence.sh <tag> [pytest] [pmc]
#   bench JSON line of the driver's command (default flags), both models' bench lines, rocprofv3 kernel stats of the timed
#   schedule (train r50 / r18, DAVIS r50), optionally the GPU test suite and the PMC traffic passes.
# Copy what should be judged from gpurun_out/ to profiles/ (tracked).
TAG=${1:-r04_x}; shift
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
for W in "$@"; do
  if [ "$W" = pytest ]; then
    timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.txt
  fi
  if [ "$W" = pmc ]; then
    for M in r50; do
      ./tools/gpu_pmc.sh $M > gpurun_out/${TAG}_pmc_$M.txt 2>&1
      python tools/make_traffic_json.py $M $TAG >> gpurun_out/${TAG}_pmc_$M.txt 2>&1; cp profiles/${TAG}_traffic_$M.json gpurun_out/
      ./tools/gpu_pmc.sh $M davis > gpurun_out/${TAG}_pmc_davis_$M.txt 2>&1
      python tools/make_traffic_json.py davis_$M $TAG >> gpurun_out/${TAG}_pmc_davis_$M.txt 2>&1; cp profiles/${TAG}_traffic_davis_$M.json gpurun_out/
    done
  fi
done
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.log
tail -4 gpurun_out/${TAG}_bench_default.log; cut -c1-600 gpurun_out/${TAG}_bench_default.json
timeout 600 python bench.py --model r18 --no-davis > gpurun_out/${TAG}_bench_r18.json 2> gpurun_out/${TAG}_bench_r18.log
tail -2 gpurun_out/${TAG}_bench_r18.log
./tools/gpu_prof.sh r50 $TAG > gpurun_out/${TAG}_prof_r50.txt 2>&1; head -14 gpurun_out/${TAG}_prof_r50.txt
./tools/gpu_prof.sh r18 $TAG > gpurun_out/${TAG}_prof_r18.txt 2>&1; head -6 gpurun_out/${TAG}_prof_r18.txt
./tools/gpu_prof.sh r50 $TAG davis > gpurun_out/${TAG}_prof_davis_r50.txt 2>&1; head -8 gpurun_out/${TAG}_prof_davis_r50.txt
