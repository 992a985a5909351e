#!/bin/bash
# GPU check of the input pipeline: parity tests + kernel timing
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pipeline.py -q -m gpu 2>&1 | tail -5
timeout 300 python tools/bench_pipeline.py 128 2>&1 | tail -2 | tee gpurun_out/pipeline_bench.json
