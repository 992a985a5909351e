#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_r18.err | tee gpurun_out/bench_r18.json | cut -c1-200; grep "timed steps" gpurun_out/bench_r18.err
./tools/gpu_prof.sh r18 prof_r18e 2>&1 | head -12
