#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_emu_train_step.py -m gpu -q -p no:cacheprovider -x -k "rccl or properties_r50_512 or r18_full" > gpurun_out/r2_pytest_gpu3.txt 2>&1; tail -4 gpurun_out/r2_pytest_gpu3.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for i in 1 2; do
for M in r50 r18; do
echo -n "$M rot=1: "; timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
echo -n "$M rot=0: "; VFS_ROT=0 timeout 300 python bench.py --model $M $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
done; done
echo -n "r50 main prio: "; VFS_MAIN_PRIO=1 timeout 300 python bench.py --model r50 $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
echo -n "r50 512: "; timeout 300 python bench.py --model r50 --size 512 $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
echo -n "r50 512 rot=0: "; VFS_ROT=0 timeout 300 python bench.py --model r50 --size 512 $B 2>&1 | grep -E "timed steps" | sed 's/.*timed steps: //'
