#!/bin/bash
# what-if builds of the halo kernel (tools/make_variant_lib.sh halo_wN conv_halo.hip -DHALO_WHATIF=N): which phase costs what
cd "$GRAFT_REPO_ROOT"
for w in ${WS:-"" 1 4 8 9 12 32}; do
  for o in ${ARMS:-halo_deep_max=256}; do
    if [ -z "$w" ]; then L=""; else L="VFS_HIP_LIB=$PWD/tools/_bin/libvfs_halo_w$w.so"; fi
    echo "== whatif [$w] $o"; env $L timeout 200 python tools/bench_halo.py 30 fd r50 $o 2>&1 | grep "^(64, 16\|^(64, 8"
  done
done
