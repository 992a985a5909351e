#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_conv -o conv -- python $GRAFT_REPO_ROOT/tools/bench_conv.py r18 > $GRAFT_REPO_ROOT/gpurun_out/prof_conv.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_conv/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
# group consecutive identical kernel names+grid into runs to show per-shape averages
agg = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name'][:60]
    key = (name, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg.setdefault(key, [0, 0]); a[0] += d; a[1] += 1
for (name, grid), (tot, n) in agg.items():
    if 'wgrad' in name or 'reduce' in name:
        print(f'{name:62s} grid {grid:>9s} calls {n:4d} avg {tot/n/1e3:8.1f} us')
PY
rm -rf gpurun_out/prof_conv
