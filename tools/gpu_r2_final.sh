#!/bin/bash
# the committed round-2 evidence: whole -m gpu suite, bench lines + rocprofv3 kernel stats + PMC traffic for both models,
# DAVIS bench lines (fp32 default, bf16 fast mode) with their kernel stats.  usage: tools/gpu_r2_final.sh <tag>
TAG=${1:-r02_b}
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.txt
./tools/gpu_profiles.sh $TAG "r50 r18" 2>&1 | grep -E "timed steps|total kernel time|^\{" | cut -c1-300
for C in "r50 fp32" "r50 bf16" "r18 fp32"; do
  set -- $C
  timeout 900 python bench.py --workload davis --model $1 --precision $2 --steps 30 --warmup 3 > gpurun_out/${TAG}_bench_davis_$1_$2.json 2> gpurun_out/${TAG}_bench_davis_$1_$2.log
  tail -2 gpurun_out/${TAG}_bench_davis_$1_$2.log | head -1; cut -c1-400 gpurun_out/${TAG}_bench_davis_$1_$2.json
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_davis -o r50 -- python $GRAFT_REPO_ROOT/bench.py --workload davis --model r50 --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_davis.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/${TAG}_prof_davis -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_davis_r50_fp32_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/${TAG}_prof_davis gpurun_out/${TAG}_prof_r50 gpurun_out/${TAG}_prof_r18
head -6 gpurun_out/${TAG}_davis_r50_fp32_kernel_stats.csv | cut -c1-200
