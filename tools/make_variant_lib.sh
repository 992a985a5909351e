#!/bin/bash
# a variant of the working tree's library with extra compiler flags on ONE source (what-if / A-B builds):
#   tools/make_variant_lib.sh NAME FILE.hip "-DHALO_WHATIF=1"   ->  tools/_bin/libvfs_NAME.so  (use with VFS_HIP_LIB=...)
# the other objects are the ones python -m vfs_amd.build left in vfs_amd/csrc/build
set -e
NAME=$1; FILE=$2; FLAGS=$3
cd "$(dirname "$0")/.."
python -m vfs_amd.build > /dev/null
mkdir -p tools/_bin /tmp/vfs_var_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $FLAGS -c vfs_amd/csrc/$FILE -o /tmp/vfs_var_$NAME/$FILE.o
OBJS=$(ls vfs_amd/csrc/build/*.o | grep -v "/$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libvfs_$NAME.so $OBJS /tmp/vfs_var_$NAME/$FILE.o
echo "tools/_bin/libvfs_$NAME.so: $FILE $FLAGS"
