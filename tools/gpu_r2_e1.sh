#!/bin/bash
# round 2, experiment batch 1: DMA ring for the statistics-fused dgrads, ring tile limit, then the per-shape kernel table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_emu_conv.py -m gpu -q -x -k "dgrad_fused or pointwise or conv_fwd_dgrad" 2>&1 | tail -3
./tools/gpu_ab_opts.sh "-" "igemm_ring_fbn=0" "igemm_ring_tiles=1024" "igemm_ring_tiles=2048,igemm_narrow_below=1025" 2>&1 | tee gpurun_out/e1_ab.txt
./tools/gpu_prof_shapes.sh r50 > gpurun_out/e1_shapes.log 2>&1
tail -5 gpurun_out/e1_shapes.log
