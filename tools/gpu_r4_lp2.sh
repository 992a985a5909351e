#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_labelprop2.py tests/test_exact_f32.py -m gpu -x -q > gpurun_out/r04_e_pytest_lp2.txt 2>&1; tail -3 gpurun_out/r04_e_pytest_lp2.txt
TAG=r04_e ./tools/gpu_davis_ab.sh "VFS_LP_TWO_PASS=0" "-"
