#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/lp2_stats.py r50 2>&1 | grep -E "two-pass|flag|bit-equal|:"
python -m pytest tests/test_emu_bn.py -m gpu -x -q 2>&1 | tail -2
