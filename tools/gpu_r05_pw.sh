#!/bin/bash
# round 5, the persistent 1x1 kernel (csrc/conv_pw.hip): bit-equality at the bench shapes, per-layer A/B, whole-step A/B - one box
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-r05_pw}
timeout 900 python -m pytest tests/test_pw.py -m gpu -x -q > gpurun_out/${TAG}_pytest.txt 2>&1; tail -3 gpurun_out/${TAG}_pytest.txt
for o in ${ARMS:-igemm_pw=0 igemm_pw=2}; do timeout 300 python tools/bench_pw.py fbn $o; done > gpurun_out/${TAG}_bench_pw.txt 2>&1
cat gpurun_out/${TAG}_bench_pw.txt
MODELS=r50 TAG=$TAG tools/gpu_ab.sh "VFS_OPTS=igemm_pw=0" "-" "VFS_OPTS=igemm_pw=2"
