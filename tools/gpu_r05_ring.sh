#!/bin/bash
# round 5: depth of the 1x1 DMA ring (PIPE 5) - 3 stages (shipped) against 4 and 5 stage what-if builds
# (tools/make_variant_lib.sh ringN conv_igemm.hip -DIGEMM_RING5=N): per-layer table, then whole-step A/B, one box
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for r in "" 4 5; do
  L=""; [ -n "$r" ] && L="VFS_HIP_LIB=$PWD/tools/_bin/libvfs_ring$r.so"
  echo "== ring ${r:-3}"; env $L timeout 300 python tools/bench_pw.py
done > gpurun_out/r05_ring_bench_pw.txt 2>&1
cat gpurun_out/r05_ring_bench_pw.txt
MODELS=r50 TAG=r05_ring tools/gpu_ab.sh "-" "VFS_HIP_LIB=$PWD/tools/_bin/libvfs_ring4.so" "VFS_HIP_LIB=$PWD/tools/_bin/libvfs_ring5.so"
