#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
./tools/gpu_ab_env.sh VFS_BNACT_FUSE 1 0 r18
./tools/gpu_ab_env.sh VFS_BNACT_FUSE 1 0 r50
