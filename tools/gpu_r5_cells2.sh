#!/bin/bash
set -u
tag=${1:-r5e3}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_cells_ab.txt
: > $out
p() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-14s %7.3f M env-steps/s  %8.2f us per step  policy kernel %7.2f us  frac %.4f' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac']))" $1; }
for i in 1 2; do
  python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | p bf16 >> $out
  python3 bench.py --workload policy_rollout --obs-dtype cells --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | p cells >> $out
  PPG_POLICY_SLOTS=0 python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | p bf16_plain >> $out
  PPG_POLICY_SLOTS=0 python3 bench.py --workload policy_rollout --obs-dtype cells --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | p cells_plain >> $out
done
cat $out
