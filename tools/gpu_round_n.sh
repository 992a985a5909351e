#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_emu_bn.py -q -m gpu -x 2>&1 | tail -2
bash tools/gpu_ab_env.sh VFS_FIN_MAX_ROWS 128 64 r50
bash tools/gpu_ab_env.sh VFS_FIN_MAX_ROWS 128 256 r50
bash tools/gpu_ab_env.sh VFS_FIN_MAX_ROWS 128 64 r18
