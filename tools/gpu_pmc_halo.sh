#!/bin/bash
# SQ counters of the 3x3 conv kernels (two passes, counters only)
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WHAT=${1:-fd}
python tools/bench_halo.py 20 $WHAT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA"
P3="GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1))
  cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmch_$i -o h -- python $GRAFT_REPO_ROOT/tools/bench_halo.py 3 $WHAT > $GRAFT_REPO_ROOT/gpurun_out/pmch_$i.log 2>&1
  echo "pass $i exit $?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for i in (1, 2, 3):
    fs = glob.glob(f'gpurun_out/pmch_{i}/*counter_collection.csv')
    if not fs:
        print('no counters for pass', i); continue
    for r in csv.DictReader(open(fs[0])):
        name = r['Kernel_Name']
        if 'conv' not in name and 'wgrad' not in name: continue
        key = (name.split('(')[0][-48:], r.get('Grid_Size', ''), r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''), r.get('LDS_Block_Size', ''))
        d = agg.setdefault(key, collections.defaultdict(lambda: [0.0, 0]))
        c = d[r['Counter_Name']]; c[0] += float(r['Counter_Value']); c[1] += 1
with open('gpurun_out/pmc_halo.txt', 'w') as f:
    for key, d in agg.items():
        line = f'{key}\n   ' + '  '.join(f'{k}={v[0] / v[1]:.4g}' for k, v in d.items())
        print(line); f.write(line + '\n')
PY
rm -rf gpurun_out/pmch_1 gpurun_out/pmch_2 gpurun_out/pmch_3
