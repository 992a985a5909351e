#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
for O in "lp2_xcd=1" "lp2_xcd=0"; do
  echo "== $O"; VFS_OPTS="$O" python tools/lp2_stats.py r50 2>&1 | grep -E "two-pass|flag|bit-equal"
  VFS_OPTS="$O" python tools/lp2_stats.py r18 2>&1 | grep -E "two-pass|dense:|flag|bit-equal"
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lp2prof -o x -- python $GRAFT_REPO_ROOT/tools/lp2_stats.py r50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/lp2prof -name "*kernel_stats.csv" | head -1); grep -E "lp2|labelprop|split_rows" $f | cut -c1-160; rm -rf gpurun_out/lp2prof
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lp2prof -o x -- python $GRAFT_REPO_ROOT/tools/lp2_stats.py r18 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/lp2prof -name "*kernel_stats.csv" | head -1); grep -E "lp2|labelprop|split_rows" $f | cut -c1-160; rm -rf gpurun_out/lp2prof
