#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/bench_davis.py --model r18 --frames 30 2>gpurun_out/davis_r18.err | tee gpurun_out/davis_r18.json; tail -3 gpurun_out/davis_r18.err
timeout 900 python tools/bench_davis.py --model r50 --frames 30 --parity-frames 2 2>gpurun_out/davis_r50.err | tee gpurun_out/davis_r50.json; tail -3 gpurun_out/davis_r50.err
