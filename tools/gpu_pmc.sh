#!/bin/bash
# HBM traffic of the bench kernels from the L2 memory-side counters (separate passes per guide)
# usage: tools/gpu_pmc.sh <r18|r50> [davis|train] [size]   (davis: the fp32 DAVIS workload instead of the train step -> gpurun_out/pmc_davis_<model>.json;
#        size 512: BASELINE configs[4] -> gpurun_out/pmc_<model>_512.json)
# Round 6: the train step is profiled on its DEFAULT schedule (command-tape replay, weight gradients on the side stream) - rounds 1-5 forced
# VFS_GRAPHS=0 VFS_SIDE_STREAM=0.  (rocprofv3 serialises the dispatches while it collects counters, so the bytes per kernel do not depend on it.)
MODEL=${1:-r18}; WORK=${2:-train}; SIZE=${3:-256}
if [ "$WORK" = davis ]; then BARGS="--workload davis --precision fp32 --steps 49 --warmup 0 --no-cpu-baseline --no-roofline"; OUT=davis_$MODEL
else BARGS="--steps 3 --warmup 1 --size $SIZE --no-cpu-baseline --no-roofline --no-davis"; OUT=$MODEL; [ "$SIZE" != 256 ] && OUT=${MODEL}_$SIZE; fi
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o $MODEL -- python $GRAFT_REPO_ROOT/bench.py --model $MODEL $BARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1
  echo "$C exit $?"
done
cd $GRAFT_REPO_ROOT
ls gpurun_out/pmc_FETCH_SIZE | head
python - $OUT <<'PY'
import csv, glob, collections, json, sys
model = sys.argv[1]
out = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    fs = glob.glob(f'gpurun_out/pmc_{C}/*counter_collection.csv')
    if not fs:
        print('no counter file for', C); continue
    rows = list(csv.DictReader(open(fs[0])))
    print(C, 'columns', list(rows[0].keys())[:14])
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        if r.get('Counter_Name') != C: continue
        name = r['Kernel_Name'].split('(')[0][:48]
        agg[name][0] += float(r['Counter_Value']); agg[name][1] += 1
    out[C] = {k: {'sum': v[0], 'calls': v[1]} for k, v in agg.items()}
json.dump(out, open(f'gpurun_out/pmc_{model}.json', 'w'), indent=1)
for k, v in sorted(out.get('FETCH_SIZE', {}).items(), key=lambda kv: -kv[1]['sum'])[:16]:
    w = out.get('WRITE_SIZE', {}).get(k, {'sum': 0, 'calls': 1})
    print(f"{k:50s} calls {v['calls']:4d} FETCH_SIZE/call {v['sum']/v['calls']:12.1f}  WRITE_SIZE/call {w['sum']/max(w['calls'],1):12.1f}")
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
