#!/bin/bash
# round 4, first call: GPU suite on the round's first code state, the evidence set, knob A/Bs for the 1x1 pipeline
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=r04_a
VFS_BENCH_SHAPES=gpurun_out/${TAG}_shapes_r50.txt timeout 600 python bench.py --no-davis --no-cpu-baseline > gpurun_out/${TAG}_bench_r50_nodavis.json 2> gpurun_out/${TAG}_bench_r50_nodavis.log
tail -2 gpurun_out/${TAG}_bench_r50_nodavis.log; head -60 gpurun_out/${TAG}_shapes_r50.txt
TAG=${TAG}_knobs MODELS=r50 ./tools/gpu_ab.sh - "VFS_OPTS=igemm_ring_tiles=4096" "VFS_OPTS=igemm_ring_tiles=1000000" "VFS_BN_FUSE=0" "VFS_OPTS=igemm_onek=1"
./tools/gpu_evidence.sh $TAG pytest
