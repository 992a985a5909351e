#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_emu_conv.py -q -m gpu -x 2>&1 | tail -3
for o in "igemm_ring_stages=3" "igemm_ring_stages=4"; do timeout 200 python tools/bench_pw.py $o 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/bench_pw.txt
bash tools/gpu_ab_env.sh VFS_OPTS igemm_ring_stages=3 igemm_ring_stages=4 r50
bash tools/gpu_ab_env.sh VFS_OPTS igemm_ring_stages=3 igemm_ring_stages=4,igemm_onek=2 r50
