#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_emu_conv.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
python tools/bench_halo.py 20 w
