#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/lp2_stats.py r50 2>&1 | grep -E "two-pass|bit-equal|dense:|^ *[0-9]+ :"
python tools/lp2_stats.py r18 2>&1 | grep -E "two-pass|bit-equal|dense:"
for O in "lp2_dbg=0" "lp2_dbg=14"; do
for M in r50 r18; do
cd /tmp && VFS_OPTS=$O timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lp2prof -o x -- python $GRAFT_REPO_ROOT/tools/lp2_stats.py $M > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/lp2prof -name "*kernel_stats.csv" | head -1); echo "$O $M: $(grep -E 'lp2_score' $f | cut -d, -f1-4)"; rm -rf gpurun_out/lp2prof
done; done
