#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_emu_train_step.py -q -m gpu -x 2>&1 | tail -5
for M in r18 r50; do timeout 300 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"; done
bash tools/gpu_trace_gaps.sh r18 | tail -8
