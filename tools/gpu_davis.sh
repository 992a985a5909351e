#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_davis_eval.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
for M in r18 r50; do timeout 600 python tools/bench_davis.py --model $M 2>&1 | tail -1 | tee gpurun_out/davis_$M.json; done
