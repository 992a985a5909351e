#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench on the schedule that bench.py TIMES (command-tape replay, weight gradients on the
# side stream): tools/gpu_prof.sh <r18|r50> <tag> [train|davis]
#   -> gpurun_out/<tag>_bench_<model>_kernel_stats.csv + .meta.json ({"passes": steps the run executed}) - copy both to profiles/
#      as rNN_bench_<model>_kernel_stats.* : bench.py reports the per-family time of the timed schedule from them.
MODEL=${1:-r50}; TAG=${2:-prof}; WORK=${3:-train}; SIZE=${4:-256}      # SIZE 512: BASELINE configs[4] -> <tag>_bench_<model>_512_kernel_stats.*
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ "$WORK" = davis ]; then
  BARGS="--workload davis --precision fp32 --steps 49 --warmup 2 --no-cpu-baseline --no-roofline"; NAME=davis_$MODEL; PASSES=140      # propagated frames: 2 warm-up + 49 untimed + 49 timed + 2 x 20 (the steady-state leg)
else
  BARGS="--steps 5 --warmup 2 --size $SIZE --no-cpu-baseline --no-roofline --no-davis"; NAME=bench_$MODEL; [ "$SIZE" != 256 ] && NAME=bench_${MODEL}_$SIZE; PASSES=9      # 2 recording passes + 2 + 5
fi
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_$NAME -o $MODEL -- python $GRAFT_REPO_ROOT/bench.py --model $MODEL $BARGS > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_$NAME.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_$NAME -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_${NAME}_kernel_stats.csv
echo "{\"passes\": $PASSES, \"command\": \"rocprofv3 --kernel-trace --stats -- python bench.py --model $MODEL $BARGS\", \"note\": \"default schedule: command-tape replay, weight gradients on the side stream; train: the two recording passes have no optimizer step\"}" > gpurun_out/${TAG}_${NAME}_kernel_stats.meta.json
python - "$f" $PASSES <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); n = float(sys.argv[2])
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot/1e6:.2f} ms over {n:.0f} passes = {tot/n/1e6:.2f} ms/pass')
for r in rows[:28]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} total {int(r['TotalDurationNs'])/n/1e6:7.3f} ms/pass avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
rm -rf gpurun_out/${TAG}_$NAME
