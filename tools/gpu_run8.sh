#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
for B in 256 384 512 768; do echo "=== halo wgrad blocks $B"; VFS_WGRAD_HALO_BLOCKS=$B timeout 300 python tools/bench_conv.py r18 2>&1 | grep -E "3, 1\)" | awk '{print $1,$2,$3,$4,$5,$6,$7, "wgrad", $10, "ms", $14, $15}'; done
