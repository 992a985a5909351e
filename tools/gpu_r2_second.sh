#!/bin/bash
# second look: the whole -m gpu suite, XCD-order A/B, batched wgrad reduction A/B, the new bench lines
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r2_pytest_gpu.txt 2>&1; tail -5 gpurun_out/r2_pytest_gpu.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for i in 1 2; do
echo -n "r50 default: "; timeout 300 python bench.py --model r50 $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
echo -n "r50 no xcd order: "; VFS_OPTS=igemm_xcd=0 timeout 300 python bench.py --model r50 $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
echo -n "r50 per-layer wgrad reduce: "; VFS_WGRAD_BATCH=0 timeout 300 python bench.py --model r50 $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done
echo -n "r18 default: "; timeout 300 python bench.py --model r18 $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
echo -n "r18 no xcd order: "; VFS_OPTS=igemm_xcd=0 timeout 300 python bench.py --model r18 $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
echo -n "r50 1-rank RCCL: "; VFS_FORCE_COLLECTIVES=1 timeout 300 python bench.py --model r50 $B 2>&1 | grep -E "timed steps|rror" | sed 's/.*timed steps: //'
timeout 600 python bench.py --model r50 --steps 20 --warmup 5 > gpurun_out/r2_bench_r50.json 2> gpurun_out/r2_bench_r50.log; tail -2 gpurun_out/r2_bench_r50.log; head -c 2500 gpurun_out/r2_bench_r50.json; echo
timeout 600 python bench.py --workload davis --model r50 --steps 20 --warmup 3 > gpurun_out/r2_bench_davis_r50.json 2> gpurun_out/r2_bench_davis_r50.log; tail -2 gpurun_out/r2_bench_davis_r50.log; head -c 2500 gpurun_out/r2_bench_davis_r50.json; echo
