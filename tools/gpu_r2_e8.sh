#!/bin/bash
# pure-GEMM prologue of the implicit-GEMM kernel: 1x1 micro-benchmark and whole step, new build vs tools/_bin/libvfs_base.so
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_emu_conv.py -m gpu -q -x 2>&1 | tail -2
echo "== new";  python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/e8_pw_new.txt
echo "== base"; VFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/_bin/libvfs_base.so python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/e8_pw_base.txt
./tools/gpu_ab_lib.sh 2>&1 | tee gpurun_out/e8_ab.txt
