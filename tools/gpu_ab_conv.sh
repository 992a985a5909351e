#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_emu_conv.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
echo "== new"; python tools/bench_conv.py ${1:-r50} 2>&1 | cut -c1-120
echo "== base"; VFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/_bin/libvfs_base.so python tools/bench_conv.py ${1:-r50} 2>&1 | cut -c1-120
