#!/bin/bash
# round-2 opening run: the whole -m gpu suite (no -x: list every failure), baselines on this box, two A/Bs
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_pytest_gpu.txt 2>&1; tail -40 gpurun_out/r2_pytest_gpu.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
for i in 1 2; do
echo -n "r50 default: "; timeout 300 python bench.py --model r50 $B 2>&1 | grep -E "timed steps"
echo -n "r50 ring_upfront: "; VFS_OPTS=igemm_ring_upfront=1 timeout 300 python bench.py --model r50 $B 2>&1 | grep -E "timed steps"
done
echo -n "r18 default: "; timeout 300 python bench.py --model r18 $B 2>&1 | grep -E "timed steps"
echo -n "r50 512: "; timeout 300 python bench.py --model r50 --size 512 $B 2>&1 | grep -E "timed steps"
echo -n "r50 1-rank RCCL: "; VFS_FORCE_COLLECTIVES=1 timeout 300 python bench.py --model r50 $B 2>&1 | grep -E "timed steps|rror"
