#!/bin/bash
# kernel statistics of the N > 1 code path on one GPU (1-rank group) beside the plain step: which launches does the path add?
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/gpu_prof.sh r50 r06_plain > gpurun_out/r06_plain_prof_r50.txt 2>&1
VFS_FORCE_COLLECTIVES=1 VFS_SYNCBN_P2P=force VFS_FIN_XCHG=0 ./tools/gpu_prof.sh r50 r06_coll > gpurun_out/r06_coll_prof_r50.txt 2>&1
python - <<'PY'
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']), int(r['TotalDurationNs']))
    return d
a, b = load('gpurun_out/r06_plain_bench_r50_kernel_stats.csv'), load('gpurun_out/r06_coll_bench_r50_kernel_stats.csv')
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    rows.append(((tb - ta) / 9e3, k, ca / 9, cb / 9, ta / 9e3, tb / 9e3))
print('kernel time per pass (us): collectives path minus plain path, largest differences')
for d, k, ca, cb, ta, tb in sorted(rows, key=lambda r: -abs(r[0]))[:40]:
    print(f'{d:9.1f} us  calls {ca:6.1f} -> {cb:6.1f}   {ta:8.1f} -> {tb:8.1f} us   {k[:90]}')
print('total', sum(v[1] for v in a.values()) / 9e6, '->', sum(v[1] for v in b.values()) / 9e6, 'ms per pass')
PY
