#!/bin/bash
# A/B of builds of the library's HOST unit (policy kernels) on one box, interleaved twice: usage tools/gpu_r5_ab_libs.sh <tag> <name> [<name> ...]
# (names of tools/_build/libppg_hip_<name>.so; "base" = the product library)
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/${tag}_ab.txt
for rep in 1 2; do
  for name in "$@"; do
    if [ "$name" = base ]; then unset PPG_HIP_LIB; else export PPG_HIP_LIB=$GRAFT_REPO_ROOT/tools/_build/libppg_hip_$name.so; fi
    python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>> gpurun_out/${tag}_bench.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep  %-16s ms_per_step %.5f  policy kernel_ms %.5f  %.2f M env-steps/s  frac %.4f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'] / 1e6, d['roofline']['frac']))" >> gpurun_out/${tag}_ab.txt
  done
done
unset PPG_HIP_LIB
cat gpurun_out/${tag}_ab.txt
tail -3 gpurun_out/${tag}_bench.err
