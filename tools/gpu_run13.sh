#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_r18.err | tee gpurun_out/bench_r18.json | cut -c1-200; grep "timed steps" gpurun_out/bench_r18.err
timeout 300 python bench.py --model r50 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_r50.err | tee gpurun_out/bench_r50.json | cut -c1-200; grep "timed steps" gpurun_out/bench_r50.err
./tools/gpu_prof.sh r50 prof_r50b 2>&1 | head -16
