#!/bin/bash
# the working tree's library with ONE source taken from another revision (which change moved the step?):
#   tools/make_mixed_lib.sh NAME REV FILE.hip  ->  tools/_bin/libvfs_NAME.so
set -e
NAME=$1; REV=$2; FILE=$3
cd "$(dirname "$0")/.."
python -m vfs_amd.build > /dev/null
rm -rf /tmp/vfs_mix_$NAME && mkdir -p /tmp/vfs_mix_$NAME/vfs_amd/csrc /tmp/vfs_mix_$NAME/include tools/_bin
cp vfs_amd/csrc/*.h /tmp/vfs_mix_$NAME/vfs_amd/csrc/; cp include/*.h /tmp/vfs_mix_$NAME/include/
git show $REV:vfs_amd/csrc/$FILE > /tmp/vfs_mix_$NAME/vfs_amd/csrc/$FILE
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c /tmp/vfs_mix_$NAME/vfs_amd/csrc/$FILE -o /tmp/vfs_mix_$NAME/$FILE.o
OBJS=$(ls vfs_amd/csrc/build/*.o | grep -v "/$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libvfs_$NAME.so $OBJS /tmp/vfs_mix_$NAME/$FILE.o
echo "tools/_bin/libvfs_$NAME.so: $FILE from $REV"
