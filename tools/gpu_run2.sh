#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 500 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r18.json 2> gpurun_out/bench_r18.err
echo "bench exit $?" >> gpurun_out/bench_r18.err
cat gpurun_out/bench_r18.json; grep bench gpurun_out/bench_r18.err | tail -12
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r18 -o r18 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_r18.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_r18 | head -20
f=$(ls gpurun_out/prof_r18/*/*kernel_stats.csv gpurun_out/prof_r18/*kernel_stats.csv 2>/dev/null | head -1); echo "stats file: $f"; head -40 "$f"
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "labelprop or seg_post or golden" > gpurun_out/pytest_lp.txt 2>&1; tail -8 gpurun_out/pytest_lp.txt
