#!/bin/bash
# Round 6, call B: per-class split plans of the generic weight gradient (A/B), SQ counters of the 1x1 implicit-GEMM kernels on HEAD
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=r06_wgrad_plan MODELS=r50 STEPS=30 ./tools/gpu_ab.sh - "VFS_WGRAD_TBG_SMALL=512" "VFS_WGRAD_TBG_SMALL=1024" "VFS_WGRAD_TBG_DEEP=128" "VFS_WGRAD_TBG_SMALL=512 VFS_WGRAD_TBG_DEEP=128" "VFS_WGRAD_TB=128"
./tools/gpu_pmc_sq.sh "python tools/bench_pw.py fbn" conv_igemm r06_sq_igemm > gpurun_out/r06_sq_igemm.log 2>&1; tail -30 gpurun_out/r06_sq_igemm.log | cut -c1-400
