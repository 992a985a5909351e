#!/bin/bash
# Round 6: Linear + BatchNorm1d + ReLU of the head in one launch, BatchNorm1d backward without its reduction launch: parity + A/B
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_emu_bn.py tests/test_emu_train_step.py tests/test_cfg1_golden.py -m gpu -x -q > gpurun_out/r06_head_pytest.txt 2>&1; grep -n "passed\|failed" gpurun_out/r06_head_pytest.txt
TAG=r06_head_fuse MODELS="r50 r18" STEPS=30 ./tools/gpu_ab.sh - "VFS_HEAD_FUSE=0"
