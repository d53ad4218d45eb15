#!/bin/bash
# On the GPU box: policy rollout with the step on the 64-register cooperative kernel (ppgch_step_q2, bfloat16 rows) against the
# same library with PPG_COOP_NO_HIGH_OCCUPANCY=1 (ppgc_step_q2), alternating processes.
set -u
tag=${1:-r5h}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_coop_high_ab.txt
: > $out
p() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-8s %7.3f M env-steps/s  %8.2f us per step  policy kernel %7.2f us  frac %.4f' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac']))" $1; }
python3 -c "
import torch
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env
e = BatchedPredPreyGrass(config_env, batch_size=4096, device='cuda:0', obs_dtype=torch.bfloat16)
print('# bfloat16 rows, 4096 envs:', e.step_kernel_name(), e.wave_plan())" 2>/dev/null >> $out
for i in 1 2 3; do
  python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | p high >> $out
  PPG_COOP_NO_HIGH_OCCUPANCY=1 python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | p normal >> $out
done
cat $out
