#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_emu_conv.py -q -m gpu -x 2>&1 | tail -3
bash tools/gpu_ab_env.sh VFS_OPTS wgrad_sb=0 wgrad_sb=1 r50
bash tools/gpu_ab_env.sh VFS_OPTS wgrad_sb=0 wgrad_sb=2 r18
bash tools/gpu_ab_env.sh VFS_WGRAD_TBG 1024 2048 r50
bash tools/gpu_ab_env.sh VFS_WGRAD_TBG 1024 512 r50
