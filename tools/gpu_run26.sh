#!/bin/bash
cd $GRAFT_REPO_ROOT
for TB in 256 512 768 1024; do echo "== target blocks $TB"; VFS_WGRAD_TB=$TB python tools/bench_halo.py 20 w; done
