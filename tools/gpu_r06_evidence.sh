#!/bin/bash
# The committed evidence of round 6 on the final code, one call: tools/gpu_r06_evidence.sh [pytest] [pmc]
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=r06_z
for W in "$@"; do
  if [ "$W" = pytest ]; then
    timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.txt
  fi
  if [ "$W" = pmc ]; then
    ./tools/gpu_pmc.sh r50 > gpurun_out/${TAG}_pmc_r50.txt 2>&1; python tools/make_traffic_json.py r50 r06 >> gpurun_out/${TAG}_pmc_r50.txt 2>&1; cp profiles/r06_traffic_r50.json gpurun_out/
    ./tools/gpu_pmc.sh r50 train 512 > gpurun_out/${TAG}_pmc_r50_512.txt 2>&1; python tools/make_traffic_json.py r50_512 r06 >> gpurun_out/${TAG}_pmc_r50_512.txt 2>&1; cp profiles/r06_traffic_r50_512.json gpurun_out/
    ./tools/gpu_pmc.sh r18 > gpurun_out/${TAG}_pmc_r18.txt 2>&1; python tools/make_traffic_json.py r18 r06 >> gpurun_out/${TAG}_pmc_r18.txt 2>&1; cp profiles/r06_traffic_r18.json gpurun_out/
    ./tools/gpu_pmc.sh r50 davis > gpurun_out/${TAG}_pmc_davis_r50.txt 2>&1; python tools/make_traffic_json.py davis_r50 r06 >> gpurun_out/${TAG}_pmc_davis_r50.txt 2>&1; cp profiles/r06_traffic_davis_r50.json gpurun_out/
    tail -2 gpurun_out/${TAG}_pmc_r50.txt | cut -c1-300
  fi
done
./tools/gpu_prof.sh r50 $TAG > gpurun_out/${TAG}_prof_r50.txt 2>&1; head -12 gpurun_out/${TAG}_prof_r50.txt
./tools/gpu_prof.sh r18 $TAG > gpurun_out/${TAG}_prof_r18.txt 2>&1; head -4 gpurun_out/${TAG}_prof_r18.txt
./tools/gpu_prof.sh r50 $TAG train 512 > gpurun_out/${TAG}_prof_r50_512.txt 2>&1; head -8 gpurun_out/${TAG}_prof_r50_512.txt
./tools/gpu_prof.sh r50 $TAG davis > gpurun_out/${TAG}_prof_davis_r50.txt 2>&1; head -6 gpurun_out/${TAG}_prof_davis_r50.txt
# the bench line last: it picks up the kernel statistics / traffic files of this call when they are copied to profiles/ (the committed line is
# re-run by the driver; this one is the builder's copy)
for f in gpurun_out/${TAG}_bench_*_kernel_stats.csv gpurun_out/${TAG}_bench_*_kernel_stats.meta.json gpurun_out/${TAG}_davis_*_kernel_stats.*; do
  [ -f "$f" ] && cp "$f" profiles/$(basename "$f" | sed "s/^${TAG}_/r06_/")
done
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.log
tail -6 gpurun_out/${TAG}_bench_default.log; cut -c1-400 gpurun_out/${TAG}_bench_default.json
VFS_BENCH_SHAPES=gpurun_out/${TAG}_per_launch_shapes_r50.txt timeout 600 python bench.py --model r50 --no-davis --no-cpu-baseline > /dev/null 2>&1
python tools/gap_table.py gpurun_out/${TAG}_per_launch_shapes_r50.txt > gpurun_out/${TAG}_gap_table_r50.txt 2>&1; tail -12 gpurun_out/${TAG}_gap_table_r50.txt
