#!/bin/bash
# round 2, experiment batch 4: linear-address 1x1 weight gradient (new) vs the generic path (option wgrad_lin=0), kernel alone and whole step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_emu_conv.py -m gpu -q -x -k "pixel_step or conv_fwd_dgrad" 2>&1 | tail -2
echo "== lin"; timeout 300 python tools/bench_wgrad.py 256 512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/e4_lin.txt
echo "== generic"; timeout 300 python tools/bench_wgrad.py 256 wgrad_lin=0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/e4_gen.txt
./tools/gpu_ab_opts.sh "-" "wgrad_lin=0" 2>&1 | tee gpurun_out/e4_ab.txt
