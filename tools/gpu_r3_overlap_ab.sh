#!/bin/bash
# DAVIS: backbone batches on their own stream, overlapped with the propagation (VFS_DAVIS_OVERLAP, default on) vs one stream
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-r03_ov}
timeout 200 python -m pytest tests/test_exact_f32.py -m gpu -q -x -k "forward_test or save_np" 2>&1 | tail -2
{
for i in 1 2; do for C in "r50 fp32" "r50 bf16" "r18 fp32"; do for O in 1 0; do
  M=${C% *}; P=${C#* }
  echo -n "$M $P overlap=$O: "; VFS_DAVIS_OVERLAP=$O timeout 300 python bench.py --workload davis --model $M --precision $P --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "ms/frame" | sed 's/.*frames: //'
done; done; done
} 2>&1 | tee gpurun_out/${TAG}_davis_overlap_ab.txt
