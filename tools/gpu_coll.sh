#!/bin/bash
# single-GPU view of the multi-GPU code path: eager step + RCCL calls in a 1-rank group
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== graphs (N=1 default)"; timeout 300 python bench.py --model r18 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
echo "== eager, no collectives"; VFS_GRAPHS=0 timeout 300 python bench.py --model r18 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
echo "== eager + collectives in a 1-rank RCCL group (the N>1 code path)"; VFS_FORCE_COLLECTIVES=1 timeout 300 python bench.py --model r18 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps|rror"
echo "== same, R50"; VFS_FORCE_COLLECTIVES=1 timeout 300 python bench.py --model r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps|rror"
