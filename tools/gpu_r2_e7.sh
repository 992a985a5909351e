#!/bin/bash
# per-(kernel, grid) durations with the bit-packed mask on / off
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
VFS_MASK_BITS=1 ./tools/gpu_prof_shapes.sh r50 > /dev/null 2>&1; cp gpurun_out/kernel_by_shape_r50.txt gpurun_out/e7_bits1.txt
VFS_MASK_BITS=0 ./tools/gpu_prof_shapes.sh r50 > /dev/null 2>&1; cp gpurun_out/kernel_by_shape_r50.txt gpurun_out/e7_bits0.txt
VFS_DEBUG_NOMASK=1 ./tools/gpu_prof_shapes.sh r50 > /dev/null 2>&1; cp gpurun_out/kernel_by_shape_r50.txt gpurun_out/e7_nomask.txt
head -1 gpurun_out/e7_bits1.txt gpurun_out/e7_bits0.txt gpurun_out/e7_nomask.txt
