#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for i in 1 2; do for V in "128 fin_fat_target=1024" "256 fin_fat_target=1024" "256 fin_fat_target=512" "512 fin_fat_target=512"; do set -- $V
  echo -n "rows=$1 $2 r50: "; VFS_FIN_MAX_ROWS=$1 VFS_OPTS=$2 timeout 300 python bench.py --model r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
done; done
for i in 1 2; do for V in "128 fin_fat_target=1024" "256 fin_fat_target=1024" "256 fin_fat_target=512"; do set -- $V
  echo -n "rows=$1 $2 r18: "; VFS_FIN_MAX_ROWS=$1 VFS_OPTS=$2 timeout 300 python bench.py --model r18 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed steps"
done; done
