#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== baseline"; python tools/bench_halo.py 20 fd
for E in 1 3 4 8 15; do echo "== HALO_EXP=$E"; VFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/_bin/libvfs_exp$E.so python tools/bench_halo.py 20 fd; done
