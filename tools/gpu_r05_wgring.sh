#!/bin/bash
# round 5, LDS-DMA ring of the 1x1 weight gradient: parity on the GPU, per-layer table per arm, whole-step A/B - one box
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-r05_wgring}
timeout 900 python -m pytest tests/test_emu_conv.py -m gpu -x -q -k "wgrad" > gpurun_out/${TAG}_pytest.txt 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_pytest.txt | tail -2
for o in ${ARMS:-wgrad_ring=0 wgrad_ring=1}; do echo "== $o"; timeout 300 python tools/bench_wgrad.py ${TBS:-256} $o 2>&1 | grep "^tb"; done > gpurun_out/${TAG}_bench.txt 2>&1
cat gpurun_out/${TAG}_bench.txt
TAG=$TAG tools/gpu_ab.sh "VFS_OPTS=wgrad_ring=0" "-"
