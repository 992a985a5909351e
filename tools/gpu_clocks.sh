#!/bin/bash
# clocks / power while the train step and the DAVIS propagation run (is the chip clock- or power-limited under these kernels?)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -vE "^=|^$" | head -20
for W in "--steps 300 --warmup 5 --no-cpu-baseline --no-roofline --no-davis" "--workload davis --steps 100 --warmup 3 --no-cpu-baseline --no-roofline"; do
  python bench.py $W > /dev/null 2> gpurun_out/clk_bench.log &
  PID=$!
  sleep 12
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.7; done
  wait $PID
  grep -E "timed steps|ms/frame" gpurun_out/clk_bench.log
done
