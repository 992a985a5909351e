#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_pytest_gpu4.txt 2>&1; tail -6 gpurun_out/r2_pytest_gpu4.txt
./tools/gpu_profiles.sh r02_a r50 2>&1 | tail -30
