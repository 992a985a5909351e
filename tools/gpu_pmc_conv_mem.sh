#!/bin/bash
# HBM-side traffic + L2 hit rate of the 3x3 conv kernels (separate passes; counters only)
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WHAT=${1:-fd}
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do i=$((i+1))
  cd /tmp && VFS_OPT_C64=${VFS_OPT_C64:-1} timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcm_$i -o h -- python $GRAFT_REPO_ROOT/tools/bench_halo.py 3 $WHAT > $GRAFT_REPO_ROOT/gpurun_out/pmcm_$i.log 2>&1
  echo "pass $i ($P) exit $?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for i in (1, 2, 3, 4):
    fs = glob.glob(f'gpurun_out/pmcm_{i}/*counter_collection.csv')
    if not fs:
        print('no counters for pass', i); continue
    for r in csv.DictReader(open(fs[0])):
        name = r['Kernel_Name']
        if 'conv' not in name and 'wgrad' not in name: continue
        key = (name.split('(')[0][-44:], r.get('Grid_Size', ''))
        d = agg.setdefault(key, collections.defaultdict(lambda: [0.0, 0]))
        c = d[r['Counter_Name']]; c[0] += float(r['Counter_Value']); c[1] += 1
for key, d in agg.items():
    print(key, '  '.join(f'{k}={v[0] / v[1]:.4g}' for k, v in d.items()))
PY
rm -rf gpurun_out/pmcm_*
