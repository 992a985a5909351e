#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_r18.err | tee gpurun_out/bench_r18.json | cut -c1-250; grep -E "steps" gpurun_out/bench_r18.err | tail -2
VFS_GRAPHS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
timeout 300 python bench.py --model r50 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_r50.err | tee gpurun_out/bench_r50.json | cut -c1-250; grep -E "steps" gpurun_out/bench_r50.err | tail -2
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
