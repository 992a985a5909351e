#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
for O in "lp2_xcd=0" "lp2_xcd=1" "lp2_xcd=2" "lp2_xcd=3"; do
  echo "== $O: $(VFS_OPTS=$O python tools/lp2_stats.py r50 2>&1 | grep -E 'two-pass')"
  cd /tmp && VFS_OPTS="$O" timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lp2pmc -o x -- python $GRAFT_REPO_ROOT/tools/lp2_stats.py r50 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  python - "$O" <<'PY'
import csv, glob, collections, sys
f = glob.glob('gpurun_out/lp2pmc/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r.get('Counter_Name') != 'FETCH_SIZE': continue
    n = r['Kernel_Name'].split('(')[0][:40]
    agg[n][0] += float(r['Counter_Value']); agg[n][1] += 1
for k, v in agg.items():
    if 'lp2_score' in k:
        print(sys.argv[1], k, 'calls', v[1], 'fetched GB per launch (average over the run)', 2 * v[0] * 1024 / v[1] / 1e9)
PY
  rm -rf gpurun_out/lp2pmc
done
