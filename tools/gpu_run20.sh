#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_emu_conv.py tests/test_emu_bn.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
python tools/bench_halo.py 20 fd
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
timeout 300 python bench.py --model r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
