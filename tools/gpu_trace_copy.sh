#!/bin/bash
# which kernels sit next to the __amd_rocclr_copyBuffer launches of a train step?  (kernel trace, eager launches)
MODEL=${1:-r18}
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && VFS_GRAPHS=${VFS_GRAPHS:-0} timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -o s -- python $GRAFT_REPO_ROOT/bench.py --model $MODEL --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/trace_copy.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0][-60:] for r in rows]
ctx = collections.Counter()
for i, n in enumerate(names):
    if 'copyBuffer' in n:
        ctx[(names[i - 1] if i else '-', names[i + 1] if i + 1 < len(names) else '-', rows[i].get('Stream_Id', ''), rows[i].get('Grid_Size_X', rows[i].get('Grid_Size', '')))] += 1
out = open('gpurun_out/trace_copy.txt', 'w')
for k, v in ctx.most_common(40):
    out.write(f'{v:4d}  prev={k[0]}  next={k[1]}  stream={k[2]} grid={k[3]}\n')
out.close()
print(open('gpurun_out/trace_copy.txt').read())
PY
