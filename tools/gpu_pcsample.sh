#!/bin/bash
# PC sampling (beta) of one command: where do the waves of a kernel spend their time?  usage: tools/gpu_pcsample.sh "<command>" <tag> [method] [unit] [interval]
CMD="$1"; TAG="${2:-pcs}"; METHOD="${3:-host_trap}"; UNIT="${4:-time}"; INTERVAL="${5:-1}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
(cd /tmp && timeout 180 rocprofv3 --pc-sampling-beta-enabled 1 --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INTERVAL --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -o s -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/$TAG.log 2>&1)
echo "exit $?"; tail -5 gpurun_out/$TAG.log
ls -la gpurun_out/$TAG/* | head
for f in gpurun_out/$TAG/*pc_sampling*.csv; do echo $f; head -3 $f | cut -c1-400; wc -l $f; done
