#!/bin/bash
# per-launch durations of the label propagation kernels over one DAVIS bench clip (kernel trace): how the time of a propagated
# frame grows with its number of key frames.  usage: tools/gpu_r3_lp_trace.sh <tag> [VFS_OPTS]
TAG=${1:-r03_lptrace}; OPTS=${2:-}
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && VFS_OPTS="$OPTS" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -o lp -- python $GRAFT_REPO_ROOT/bench.py --workload davis --model r50 --precision fp32 --steps 30 --warmup 0 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/$TAG -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/${TAG}_per_launch.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
byk = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    byk[n.split('(')[0][:60]].append((d, r.get('Grid_Size_Y', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', '')))
for n, v in byk.items():
    if 'labelprop_f32_kernel' in n:
        last = v[-30:]            # the timed pass (an untimed pass over the same clip precedes it)
        print(n, 'launches', len(v))
        print(' '.join(f'{d:.0f}[{g}]' for d, g, _ in last))
tot = collections.Counter()
cnt = collections.Counter()
for n, v in byk.items():
    tot[n] = sum(d for d, _, _ in v); cnt[n] = len(v)
for n, t in tot.most_common(12):
    print(f'{n:60s} calls {cnt[n]:5d} total {t/1e3:9.2f} ms avg {t/cnt[n]:9.1f} us')
PY
rm -rf gpurun_out/$TAG
