#!/bin/bash
# round 2, experiment batch 2: paired prefetch in the generic weight-gradient kernel vs tools/_bin/libvfs_base.so
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_emu_conv.py -m gpu -q -x -k "pixel_step or conv_fwd_dgrad or folded" 2>&1 | tail -3
./tools/gpu_ab_lib.sh 2>&1 | tee gpurun_out/e2_ab.txt
./tools/gpu_prof_shapes.sh r50 > gpurun_out/e2_shapes.log 2>&1
grep -E "conv_wgrad_kernel" gpurun_out/kernel_by_shape_r50.txt | head
