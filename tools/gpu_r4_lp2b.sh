#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/lp2_stats.py r50 2>&1 | grep -E "two-pass|flag|bit-equal"
python tools/lp2_stats.py r18 2>&1 | grep -E "two-pass|dense:|bit-equal"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lp2prof -o x -- python $GRAFT_REPO_ROOT/tools/lp2_stats.py r50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/lp2prof -name "*kernel_stats.csv" | head -1); grep -E "lp2|split_rows" $f | cut -c1-120; rm -rf gpurun_out/lp2prof
timeout 900 python -m pytest tests/test_labelprop2.py tests/test_exact_f32.py tests/test_davis_eval.py -m gpu -x -q > gpurun_out/r04_b_pytest_lp2.txt 2>&1; tail -3 gpurun_out/r04_b_pytest_lp2.txt
TAG=r04_b ./tools/gpu_davis_ab.sh "VFS_LP_TWO_PASS=0" "-"
