#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
timeout 300 python bench.py --model r18 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
timeout 300 python bench.py --model r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
./tools/gpu_prof_shapes.sh ${1:-r18} | head -${2:-60}
