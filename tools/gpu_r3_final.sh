#!/bin/bash
# round 3, last GPU call: the whole GPU suite on the final code, the driver's bench line (R50 train step + DAVIS leg), the R18 line,
# kernel statistics of the DAVIS workload, the staging probe's summary.  usage: tools/gpu_r3_final.sh <tag>
TAG=${1:-r03_z}
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.txt
tail -2 gpurun_out/${TAG}_pytest_gpu.txt
timeout 200 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.log; tail -2 gpurun_out/${TAG}_bench_default.log | cut -c1-300
timeout 100 python bench.py --model r18 --no-davis > gpurun_out/${TAG}_bench_r18.json 2> gpurun_out/${TAG}_bench_r18.log; tail -1 gpurun_out/${TAG}_bench_r18.log | cut -c1-300
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_davis -o davis -- python $GRAFT_REPO_ROOT/bench.py --workload davis --model r50 --precision fp32 --steps 30 --warmup 0 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_davis.log 2>&1)
cp $(find gpurun_out/${TAG}_prof_davis -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_davis_r50_fp32_kernel_stats.csv; rm -rf gpurun_out/${TAG}_prof_davis
head -6 gpurun_out/${TAG}_davis_r50_fp32_kernel_stats.csv | cut -c1-160
timeout 60 tools/_build/probe_lp_stage summary > gpurun_out/${TAG}_probe_lp_stage_summary.txt 2>&1; tail -4 gpurun_out/${TAG}_probe_lp_stage_summary.txt
