#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_emu_conv.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
python tools/bench_halo.py 20 ${1:-fd}
