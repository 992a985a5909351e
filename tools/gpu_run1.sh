#!/bin/bash
# first GPU session: probes, parity tests, bench, rocprof
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
nproc > gpurun_out/host.txt; free -g >> gpurun_out/host.txt; rocm-smi --showproductname >> gpurun_out/host.txt 2>&1
./tools/probe_gpu.bin > gpurun_out/probe.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.txt 2>&1
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r18.json 2> gpurun_out/bench_r18.err
echo "bench exit $?" >> gpurun_out/bench_r18.err
tail -3 gpurun_out/pytest_gpu.txt; cat gpurun_out/smoke.txt | tail -3; cat gpurun_out/bench_r18.json; tail -5 gpurun_out/bench_r18.err
