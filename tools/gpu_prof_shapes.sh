#!/bin/bash
# per-(kernel, grid) average durations of one bench run: which layer shapes cost what
MODEL=${1:-r18}
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && VFS_GRAPHS=0 VFS_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_shapes -o s -- python $GRAFT_REPO_ROOT/bench.py --model $MODEL --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_shapes.log 2>&1
cd $GRAFT_REPO_ROOT
python - $MODEL <<'PY'
import csv, glob, collections, sys
f = glob.glob('gpurun_out/prof_shapes/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name'].split('(')[0][-52:]
    key = (name, r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', ''))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg.setdefault(key, [0, 0]); a[0] += d; a[1] += 1
steps = 6
items = sorted(agg.items(), key=lambda kv: -kv[1][0])
with open(f'gpurun_out/kernel_by_shape_{sys.argv[1]}.txt', 'w') as out:
    tot = sum(v[0] for v in agg.values())
    out.write(f'total kernel time per step {tot / steps / 1e6:.3f} ms (6 steps incl. warmup)\n')
    for (name, gx, gy), (t, n) in items:
        out.write(f'{name:54s} grid {gx:>9s} x{gy:>4s} calls/step {n / steps:6.1f} avg {t / n / 1e3:8.1f} us  per-step {t / steps / 1e3:8.1f} us\n')
print(open(f'gpurun_out/kernel_by_shape_{sys.argv[1]}.txt').read()[:9000])
PY
rm -rf gpurun_out/prof_shapes
