#!/bin/bash
# round 5, pass 1 of the two-pass label propagation: parity on the GPU, then the DAVIS leg per arm (each argument = environment
# assignments, "-" = the shipped build), two interleaved rounds; the per-kernel split of one arm by rocprofv3 with PROF=1
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-r05_lp2}
timeout 900 python -m pytest tests/test_labelprop2.py -m gpu -x -q > gpurun_out/${TAG}_pytest.txt 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_pytest.txt | tail -2
B="--model ${MODEL:-r50} --workload davis --precision fp32 --steps ${STEPS:-49} --warmup 2 --no-cpu-baseline --no-roofline"
{
for i in 1 2; do for E in "$@"; do
  if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
  echo -n "[$E]: "; env $EE timeout 300 python bench.py $B 2>&1 | grep -E "^\{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step'), 'ms/frame; steady', d.get('steady_state_ms_per_frame'))"
done; done
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
if [ -n "$PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o p -- python $GRAFT_REPO_ROOT/bench.py $B > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MaxNs'])
PY
fi
