#!/bin/bash
# round 5, the deep schedule of the 3x3 halo kernel: parity on the GPU, per-layer A/B, whole-step A/B - one box
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-r05_halo}
timeout 900 python -m pytest tests/test_emu_conv.py -m gpu -x -q > gpurun_out/${TAG}_pytest.txt 2>&1; tail -3 gpurun_out/${TAG}_pytest.txt
for o in halo_deep_max=0 halo_deep_max=256 halo_deep_max=512; do timeout 300 python tools/bench_halo.py 30 fd r50 $o; done > gpurun_out/${TAG}_bench_halo.txt 2>&1
cat gpurun_out/${TAG}_bench_halo.txt
MODELS=r50 TAG=$TAG tools/gpu_ab.sh "VFS_OPTS=igemm_pw=0,halo_deep_max=0" "VFS_OPTS=igemm_pw=0" "VFS_OPTS=igemm_pw=0,halo_deep_max=512"
