#!/bin/bash
# Round 6, in-launch split-K reduction of the weight gradients (vfs_conv_wgrad_inl): GPU parity + whole-step A/B + kernel stats.
#   tools/gpu_r06_wgrad.sh [tag] [pytest]
TAG=${1:-r06_wgrad}
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/_bin/probe_dma_oob > gpurun_out/${TAG}_probe_dma_oob.txt 2>&1; cat gpurun_out/${TAG}_probe_dma_oob.txt
if [ "$2" = pytest ]; then
  timeout 1200 python -m pytest tests/test_emu_conv.py -m gpu -x -q > gpurun_out/${TAG}_pytest.txt 2>&1; tail -4 gpurun_out/${TAG}_pytest.txt
fi
TAG=$TAG STEPS=30 ./tools/gpu_ab.sh - "VFS_WGRAD_INL=0"
./tools/gpu_prof.sh r50 $TAG > gpurun_out/${TAG}_prof_r50.txt 2>&1; head -40 gpurun_out/${TAG}_prof_r50.txt
