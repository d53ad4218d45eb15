#!/bin/bash
# On the GPU box: rows in the cell layout (obs_dtype 3) -- parity test, then policy rollout bf16 rows vs cells, alternating
set -u
tag=${1:-r5e}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_cells_ab.txt
timeout 600 python3 -m pytest tests/test_policy.py -m gpu -x -q -k "cell_layout" 2>&1 | grep -v "^$" | tail -12 > $out
p() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-6s %7.3f M env-steps/s  %8.2f us per step  policy kernel %7.2f us  frac %.4f  agents per env %.2f' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], d['config']['mean_agents_per_env']))" $1; }
for i in 1 2 3; do
  python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>>gpurun_out/${tag}_err.txt | p bf16 >> $out
  python3 bench.py --workload policy_rollout --obs-dtype cells --steps 100 --warmup 10 --no-cpu-baseline 2>>gpurun_out/${tag}_err.txt | p cells >> $out
done
cat $out; tail -5 gpurun_out/${tag}_err.txt | grep -v amdgpu
