#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite (new: per-stage oracle comparison at the bench sizes, 224x224 goldens,
# > 21-frame 480x854 clips, pairwise CosineSimLoss), then the default bench line (train leg + DAVIS leg).
# usage: tools/gpu_r3_first.sh <tag>
TAG=${1:-r03_a}
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; tail -25 gpurun_out/${TAG}_pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/${TAG}_bench_r50.json 2> gpurun_out/${TAG}_bench_r50.log
grep -E "timed steps|ms/frame|eager steps" gpurun_out/${TAG}_bench_r50.log; cut -c1-600 gpurun_out/${TAG}_bench_r50.json
