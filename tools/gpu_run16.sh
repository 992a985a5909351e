#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
for SS in 1 0; do echo "== VFS_SIDE_STREAM=$SS"
VFS_SIDE_STREAM=$SS timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
VFS_SIDE_STREAM=$SS VFS_GRAPHS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
VFS_SIDE_STREAM=$SS timeout 300 python bench.py --model r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
done
