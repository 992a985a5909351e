#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/bench_conv.py r18 2>&1 | tee gpurun_out/bench_conv_r18.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_r18.err | tee gpurun_out/bench_r18.json; grep "timed steps" gpurun_out/bench_r18.err
