#!/bin/bash
# round 3, second GPU call: the new tests only (P2P SyncBN exchange between two processes on one GPU, 224x224 goldens, R18 224 per
# stage, pairwise loss), the two-chain concurrency experiment, and the copyBuffer count at two step counts.  usage: <tag>
TAG=${1:-r03_b}
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_p2p.py tests/test_cfg1_golden.py tests/test_simloss.py tests/test_emu_train_step.py -m gpu -q -p no:cacheprovider -s \
  -k "p2p or two_processes or golden_at_224 or r18_224 or simloss or pairwise or hip_matches or eval_after" > gpurun_out/${TAG}_pytest_new.txt 2>&1; tail -30 gpurun_out/${TAG}_pytest_new.txt | cut -c1-400
timeout 600 python tools/exp_two_chains.py --model r50 > gpurun_out/${TAG}_two_chains.txt 2>&1; tail -2 gpurun_out/${TAG}_two_chains.txt
timeout 600 python tools/exp_two_chains.py --model r18 >> gpurun_out/${TAG}_two_chains.txt 2>&1; tail -1 gpurun_out/${TAG}_two_chains.txt
for K in 5 25; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_cp$K -o r50 -- python $GRAFT_REPO_ROOT/bench.py --steps $K --warmup 2 --no-cpu-baseline --no-roofline --no-davis > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_cp$K.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find gpurun_out/${TAG}_cp$K -name "*kernel_stats.csv" | head -1)
  echo "steps=$K (+2 warm-up +2 init passes):"; grep -E "copyBuffer|fillBuffer|FillFunctor|sgd_kernel" $f | cut -d, -f1-3 | cut -c1-120
  cp $f gpurun_out/${TAG}_cp${K}_kernel_stats.csv; rm -rf gpurun_out/${TAG}_cp$K
done
