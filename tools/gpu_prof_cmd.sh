#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python command: tools/gpu_prof_cmd.sh TAG script.py args...  -> per-kernel average durations
TAG=$1; shift
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python $R/"$@" > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -rf gpurun_out/prof_$TAG
