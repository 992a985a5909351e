#!/bin/bash
# Round 6, the N > 1 code path on one GPU (1-rank group, VFS_FORCE_COLLECTIVES=1): two-process GPU tests + whole-step A/B
#   tools/gpu_r06_syncbn.sh [pytest]
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$1" = pytest ]; then
  timeout 1500 python -m pytest tests/test_p2p.py tests/test_emu_bn.py tests/test_emu_train_step.py -m gpu -x -q -k "p2p or bn or rccl or two_processes" > gpurun_out/r06_syncbn_pytest.txt 2>&1; tail -4 gpurun_out/r06_syncbn_pytest.txt
fi
F="VFS_FORCE_COLLECTIVES=1 VFS_SYNCBN_P2P=force"
TAG=r06_syncbn_one_gpu_step MODELS="r50 r18" STEPS=30 ./tools/gpu_ab.sh - "$F" "$F VFS_DDP_SIDE=0" "$F VFS_FIN_XCHG=0" "$F VFS_FIN_XCHG=0 VFS_DDP_SIDE=0" "$F VFS_MAIN_PRIO=0" "VFS_FORCE_COLLECTIVES=1"
