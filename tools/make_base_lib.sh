#!/bin/bash
# build tools/_bin/libvfs_base.so from the csrc/ + include/ of a git revision (default HEAD), for tools/gpu_ab_lib.sh
REV=${1:-HEAD}
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/vfs_basebuild; rm -rf $T; mkdir -p $T
(cd $ROOT && git archive $REV vfs_amd/csrc include) | tar -x -C $T
cd $T/vfs_amd/csrc
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -c $f -o ${f%.hip}.o 2>/dev/null & done; wait
mkdir -p $ROOT/tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_bin/libvfs_base.so *.o
nm -D $ROOT/tools/_bin/libvfs_base.so | grep -c " T vfs_"
