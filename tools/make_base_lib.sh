#!/bin/bash
# the library of a git revision (default HEAD) as tools/_bin/libvfs_base.so, for same-box A/B against the working tree:
#   tools/make_base_lib.sh [rev]; then  gpurun -- 'bash tools/gpu_davis_ab.sh VFS_HIP_LIB=$PWD/tools/_bin/libvfs_base.so -'
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/.."
rm -rf /tmp/vfs_base && mkdir -p /tmp/vfs_base tools/_bin
git archive "$REV" vfs_amd/csrc include | tar -x -C /tmp/vfs_base
cd /tmp/vfs_base/vfs_amd/csrc
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c "$f" -o "$f.o" & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OLDPWD/tools/_bin/libvfs_base.so" *.o
echo "tools/_bin/libvfs_base.so = $REV"
