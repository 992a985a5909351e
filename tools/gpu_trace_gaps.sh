#!/bin/bash
# idle time between consecutive kernels of the replayed train step (kernel trace, single stream)
MODEL=${1:-r50}
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_gaps && VFS_SIDE_STREAM=${VFS_SIDE_STREAM:-0} timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gaps -o s -- python $GRAFT_REPO_ROOT/bench.py --model $MODEL --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/trace_gaps.log 2>&1
cd $GRAFT_REPO_ROOT
grep "timed steps" gpurun_out/trace_gaps.log
python - $MODEL <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/prof_gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last step: find the last sgd kernel and the one before it
idx = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
span = int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])
gaps = [int(step[i + 1]['Start_Timestamp']) - int(step[i]['End_Timestamp']) for i in range(len(step) - 1)]
pos = [g for g in gaps if g > 0]
hist = collections.Counter(min(g // 1000, 10) for g in pos)
out = open(f'gpurun_out/trace_gaps_{sys.argv[1]}.txt', 'w')
out.write(f'launches {len(step)} span {span / 1e6:.3f} ms busy {busy / 1e6:.3f} ms idle {sum(pos) / 1e6:.3f} ms overlapped {-sum(g for g in gaps if g < 0) / 1e6:.3f} ms\n')
out.write('gap histogram (us bucket: count): ' + ' '.join(f'{k}:{hist[k]}' for k in sorted(hist)) + '\n')
big = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:12]
for i in big:
    out.write(f'  gap {gaps[i] / 1e3:7.1f} us after {step[i]["Kernel_Name"].split("(")[0][-50:]} before {step[i + 1]["Kernel_Name"].split("(")[0][-50:]}\n')
out.close()
print(open(f'gpurun_out/trace_gaps_{sys.argv[1]}.txt').read())
PY
