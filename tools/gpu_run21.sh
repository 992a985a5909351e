#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== new"; python tools/bench_halo.py 30 fd
echo "== old"; VFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/_bin/libvfs_old.so python tools/bench_halo.py 30 fd
done
