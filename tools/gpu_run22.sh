#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== dma"; python tools/bench_halo.py 30 fd
echo "== reg"; VFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/_bin/libvfs_reg.so python tools/bench_halo.py 30 fd
done
