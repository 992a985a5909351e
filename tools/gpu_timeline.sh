#!/bin/bash
# per-launch timeline of ONE replayed train step (default launch mode, side stream on): start offset, duration, queue, kernel, grid
MODEL=${1:-r50}
TAG=${2:-r2}
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_tl && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o s -- python $GRAFT_REPO_ROOT/bench.py --model $MODEL --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/timeline.log 2>&1
cd $GRAFT_REPO_ROOT
grep "timed steps" gpurun_out/timeline.log
python - $MODEL $TAG <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/prof_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
qs = {}
with open(f'gpurun_out/timeline_{sys.argv[1]}_{sys.argv[2]}.txt', 'w') as out:
    out.write(f'# launches {len(step)} span {(int(step[-1]["End_Timestamp"]) - t0) / 1e6:.3f} ms; columns: start_us dur_us queue grid kernel\n')
    for r in step:
        q = qs.setdefault(r.get('Queue_Id', '?'), len(qs))
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')[-60:]
        g = r.get('Grid_Size_X', r.get('Grid_Size', '?'))
        out.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f} q{q} {g:>8s}x{r.get("Grid_Size_Y", "1"):<3s} {name}\n')
print(open(f'gpurun_out/timeline_{sys.argv[1]}_{sys.argv[2]}.txt').read()[:3000])
PY
