#!/bin/bash
# the library of a git revision (default HEAD) as tools/_bin/libvfs_<name>.so (default name: base), for same-box A/B against the
# working tree:   tools/make_base_lib.sh [rev] [name]; then  gpurun -- 'bash tools/gpu_ab.sh VFS_HIP_LIB=$PWD/tools/_bin/libvfs_base.so -'
set -e
REV=${1:-HEAD}; NAME=${2:-base}
cd "$(dirname "$0")/.."
rm -rf /tmp/vfs_$NAME && mkdir -p /tmp/vfs_$NAME tools/_bin
git archive "$REV" vfs_amd/csrc include | tar -x -C /tmp/vfs_$NAME
cd /tmp/vfs_$NAME/vfs_amd/csrc
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c "$f" -o "$f.o" 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OLDPWD/tools/_bin/libvfs_$NAME.so" *.o
echo "tools/_bin/libvfs_$NAME.so = $REV"
