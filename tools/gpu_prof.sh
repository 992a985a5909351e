#!/bin/bash
# rocprof kernel stats of the bench (args: model tag)
MODEL=${1:-r18}; TAG=${2:-prof}
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && VFS_GRAPHS=0 VFS_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -o $MODEL -- python $GRAFT_REPO_ROOT/bench.py --model $MODEL --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-davis > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot/1e6:.2f} ms over 7 steps = {tot/7e6:.2f} ms/step')
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} total {int(r['TotalDurationNs'])/7e6:7.3f} ms/step avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
rm -f gpurun_out/$TAG/*.db gpurun_out/$TAG/*kernel_trace.csv
