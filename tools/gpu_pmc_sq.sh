#!/bin/bash
# SQ counters of one micro-benchmark, per (kernel, grid): tools/gpu_pmc_sq.sh "<python command>" <kernel-substring> <tag>
# counters only (separate passes, --kernel-trace for the names); no tracing domains
CMD="$1"; FILT="${2:-conv}"; TAG="${3:-sq}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR"; do i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcsq_$i -o h -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/pmcsq_$i.log 2>&1)
  echo "pass $i ($P) exit $?"
done
python - "$FILT" "$TAG" <<'PY'
import csv, glob, collections, sys
filt, tag = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
for i in range(1, 7):
    fs = glob.glob(f'gpurun_out/pmcsq_{i}/*counter_collection.csv')
    if not fs:
        print('no counters for pass', i); continue
    for r in csv.DictReader(open(fs[0])):
        name = r['Kernel_Name']
        if filt not in name: continue
        key = (name.split('(')[0][-48:], r.get('Grid_Size', ''))
        d = agg.setdefault(key, collections.defaultdict(lambda: [0.0, 0]))
        c = d[r['Counter_Name']]; c[0] += float(r['Counter_Value']); c[1] += 1
with open(f'gpurun_out/pmc_{tag}.txt', 'w') as f:
    for key, d in agg.items():
        line = f'{key}\n   ' + '  '.join(f'{k}={v[0] / v[1]:.4g}' for k, v in d.items())
        print(line); f.write(line + '\n')
PY
rm -rf gpurun_out/pmcsq_*
