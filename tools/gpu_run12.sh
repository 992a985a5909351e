#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for F in 1 0; do
echo "=== VFS_STEM_FUSED=$F"
VFS_STEM_FUSED=$F timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep "timed steps"
VFS_STEM_FUSED=$F ./tools/gpu_prof.sh r18 prof_tmp 2>&1 | grep -E "total kernel|stem|maxpool"
done
