#!/bin/bash
# A/B on one box: the committed library vs a variant (VFS_HIP_LIB) with tools/bench_halo.py
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== new"; python tools/bench_halo.py 30 ${1:-w}
echo "== base"; VFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/_bin/libvfs_base.so python tools/bench_halo.py 30 ${1:-w}
done
