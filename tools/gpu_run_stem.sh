#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_emu_bn.py tests/test_emu_conv.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
./tools/gpu_prof_shapes.sh r18 | grep -E "total kernel|stem|maxpool"
