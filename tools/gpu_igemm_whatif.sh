#!/bin/bash
# what-if builds of the 1x1 DMA ring (tools/make_variant_lib.sh ig_wN conv_igemm.hip -DIGEMM_WHATIF=N): which phase costs what
cd "$GRAFT_REPO_ROOT"
for w in ${WS:-"" 1 2 4 8 9 15}; do
  if [ -z "$w" ]; then L=""; else L="VFS_HIP_LIB=$PWD/tools/_bin/libvfs_ig_w$w.so"; fi
  echo "== whatif [$w]"; env $L timeout 200 python tools/bench_pw.py 2>&1 | grep "^(64, 16\|^(64, 8\|^(64, 32"
done
