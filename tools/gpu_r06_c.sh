#!/bin/bash
# Round 6, call C: SQ counters of the 1x1 implicit-GEMM kernels on HEAD (micro-benchmark at the bench shapes) + the default bench line with its new legs
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/gpu_pmc_sq.sh "python $GRAFT_REPO_ROOT/tools/bench_pw.py fbn" conv_igemm r06_sq_igemm > gpurun_out/r06_sq_igemm.log 2>&1; tail -40 gpurun_out/r06_sq_igemm.log | cut -c1-600
timeout 900 python bench.py > gpurun_out/r06_c_bench_default.json 2> gpurun_out/r06_c_bench_default.log; tail -5 gpurun_out/r06_c_bench_default.log; python - <<'PY'
import json
r = json.load(open('gpurun_out/r06_c_bench_default.json'))
print(r['value'], r['ms_per_step'])
for k in ('r18', 'r50_512'):
    print(k, json.dumps(r.get(k))[:700])
PY
