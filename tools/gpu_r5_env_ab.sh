#!/bin/bash
# A/B of an environment switch of the library on one box, interleaved twice: usage tools/gpu_r5_env_ab.sh <tag> <VAR> <value> [<value> ...]
set -u
tag=$1; var=$2; shift 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/${tag}_ab.txt
for rep in 1 2; do
  for v in "$@"; do
    env $var=$v python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>> gpurun_out/${tag}_bench.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep  $var=%-6s ms_per_step %.5f  policy kernel_ms %.5f  %.2f M env-steps/s  frac %.4f' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'] / 1e6, d['roofline']['frac']))" >> gpurun_out/${tag}_ab.txt
  done
done
cat gpurun_out/${tag}_ab.txt
tail -3 gpurun_out/${tag}_bench.err
