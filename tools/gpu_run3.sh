#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r18.json 2> gpurun_out/bench_r18.err
echo "bench exit $?" >> gpurun_out/bench_r18.err
cat gpurun_out/bench_r18.json; grep bench gpurun_out/bench_r18.err | tail -6
timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>&1 | tail -2
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r18 -o r18 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_r18.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof_r18 | head -20
f=$(find gpurun_out/prof_r18 -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; head -45 "$f"
rm -f gpurun_out/prof_r18/*.db gpurun_out/prof_r18/*kernel_trace.csv
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "labelprop" > gpurun_out/pytest_lp.txt 2>&1; tail -8 gpurun_out/pytest_lp.txt
timeout 300 python bench.py --model r50 --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/bench_r50.err | tee gpurun_out/bench_r50.json; tail -3 gpurun_out/bench_r50.err
