#!/bin/bash
# does the train step run power- / clock-limited?  samples rocm-smi while bench.py runs 200 steps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(rocm-smi --showclocks --showpower --showtemp 2>&1 | head -40) > gpurun_out/clocks_idle.txt
timeout 300 python bench.py --model ${1:-r50} --steps 300 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/clocks_bench.log 2>&1 &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do (rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power|Socket" ) >> gpurun_out/clocks_load.txt; echo "--" >> gpurun_out/clocks_load.txt; sleep 0.3; done
wait $BP
grep "timed steps" gpurun_out/clocks_bench.log
grep -E "sclk|mclk|Power|Socket" gpurun_out/clocks_idle.txt | head -8; echo ==== ; head -24 gpurun_out/clocks_load.txt
