#!/bin/bash
# launch modes of the train step on one GPU: graphs / command tape / eager, without and with collectives (1-rank RCCL group = the N>1 code path)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_emu_train_step.py -q -m gpu -x 2>&1 | tail -3
for M in r50 r18; do
echo "== $M graphs"; VFS_GRAPHS=1 timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
echo "== $M tape (default)"; timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
echo "== $M eager, no collectives"; VFS_TAPE=0 timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
echo "== $M tape + collectives in a 1-rank RCCL group (the N>1 code path)"; VFS_FORCE_COLLECTIVES=1 timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps|rror"
echo "== $M eager + collectives"; VFS_TAPE=0 VFS_FORCE_COLLECTIVES=1 timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps|rror"
done
