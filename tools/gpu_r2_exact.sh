#!/bin/bash
# the fp32 evaluation path on hardware: bit-exactness tests, then DAVIS timings in both precisions
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_exact_f32.py tests/test_emu_labelprop.py -m gpu -q -p no:cacheprovider > gpurun_out/r2_pytest_exact.txt 2>&1; tail -25 gpurun_out/r2_pytest_exact.txt
for M in r18 r50; do for P in fp32 bf16; do
  timeout 600 python tools/bench_davis.py --model $M --frames 12 --parity-frames 2 --precision $P > gpurun_out/r2_davis_${M}_${P}.json 2> gpurun_out/r2_davis_${M}_${P}.err; tail -c 1500 gpurun_out/r2_davis_${M}_${P}.json; tail -3 gpurun_out/r2_davis_${M}_${P}.err
done; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
