#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== 1-rank RCCL group, collectives forced (SyncBN stat all-reduces + bucketed gradient all-reduce + log vars)"
VFS_FORCE_COLLECTIVES=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_coll.err | cut -c1-220; grep -E "timed steps|Error|error" gpurun_out/bench_coll.err | tail -3
echo "== torch.distributed.run launcher, 1 process"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_trun.err | cut -c1-160; grep -E "timed steps|Error|error" gpurun_out/bench_trun.err | tail -3
timeout 300 python bench.py --model r50 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_r50.err | tee gpurun_out/bench_r50.json | cut -c1-200; grep "timed steps" gpurun_out/bench_r50.err
