#!/bin/bash
# conv_f32 what-if table (WRONG results, timing only): average launch time of the family in the DAVIS ResNet-50 pass with parts of the
# kernel switched off (VFS_OPTS=conv_f32_dbg: 1 no gather, 2 no LDS stores, 4 no MFMAs).  Output also in gpurun_out/${TAG}_conv_whatif.txt
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
TAG=${TAG:-r04}
{
for E in "$@"; do
  if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
  echo -n "[$E]: "
  env $EE timeout 300 python bench.py --workload davis --model ${MODEL:-r50} --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l)['roofline']
        for f in [r] + r.get('families', []):
            if f['kernel'] == 'conv_f32': print('conv_f32 %.4f ms per launch, %.1f TFLOP/s, frac %.3f' % (f['avg_launch_ms'], f['achieved'], f['frac']))
"
done
} 2>&1 | tee gpurun_out/${TAG}_conv_whatif.txt
