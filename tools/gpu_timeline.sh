#!/bin/bash
# Timeline of the timed schedule: rocprofv3 --kernel-trace of a short bench run, then per step the idle time of each queue and the
# tail after the main chain's last backward kernel (tools/timeline_tail.py).  tools/gpu_timeline.sh [r50|r18] [tag]
MODEL=${1:-r50}; TAG=${2:-timeline}
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tl -o $MODEL -- python $GRAFT_REPO_ROOT/bench.py --model $MODEL --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-davis > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tl.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_tl -name "*kernel_trace.csv" | head -1)
python tools/timeline_tail.py "$f" | tee gpurun_out/${TAG}_timeline_$MODEL.txt
rm -rf gpurun_out/${TAG}_tl
