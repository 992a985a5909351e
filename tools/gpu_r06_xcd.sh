#!/bin/bash
# Round 6 EXPERIMENT: in-launch split-K reduction through ONE XCD's L2 (VFS_WGRAD_INL=2; workgroup b assumed on XCD b % 8)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
VFS_WGRAD_INL=2 timeout 1200 python -m pytest tests/test_emu_train_step.py -m gpu -x -q -k "every_stage_matches_oracle_at_bench_sizes or properties_at_baseline" > gpurun_out/r06_xcd_pytest.txt 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r06_xcd_pytest.txt | head -5
TAG=r06_wgrad_inl_same_xcd MODELS="r50 r18" STEPS=30 ./tools/gpu_ab.sh - "VFS_WGRAD_INL=2"
VFS_WGRAD_INL=2 ./tools/gpu_prof.sh r50 r06_xcd > gpurun_out/r06_xcd_prof_r50.txt 2>&1; head -24 gpurun_out/r06_xcd_prof_r50.txt
