#!/bin/bash
# round 2, experiment batch 3: the generic weight-gradient kernel alone (new vs base build, split plans) + the counter list
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== new"; timeout 300 python tools/bench_wgrad.py 256 512 128 2>&1 | tee gpurun_out/e3_new.txt
echo "== base"; VFS_HIP_LIB=$GRAFT_REPO_ROOT/tools/_bin/libvfs_base.so timeout 300 python tools/bench_wgrad.py 256 2>&1 | tee gpurun_out/e3_base.txt
(rocprofv3 -L 2>&1 || rocprofv3 --list-avail 2>&1) > gpurun_out/e3_counters.txt
grep -c . gpurun_out/e3_counters.txt
