#!/bin/bash
# On the GPU box: BASELINE config 4 (pair kernel, 21.6 KB of LDS per env = 7 envs per CU) with FEWER envs per CU (unused LDS added:
# PPG_STEP_LDS_PAD), alternating; the same for the 256-env kernel is meaningless (one env per CU).
set -u
tag=${1:-r5g}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_c4_occupancy.txt
lds=$(python3 -c "
import torch
from predpreygrass_amd.batched import BatchedPredPreyGrass
from predpreygrass_amd.config import config_env
e = BatchedPredPreyGrass({**config_env, 'grid_size': 64, 'n_initial_active_predator': 16, 'n_initial_active_prey': 32, 'predator_obs_range': 7, 'prey_obs_range': 7}, batch_size=4096, device='cuda:0')
print(e._lib.ppg_lds_bytes(e._handle))" 2>/dev/null | tail -1)
echo "# pair kernel, LDS per env $lds bytes" > $out
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('envs per CU %-3s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  %s' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], r.get('kernel')))" "$1"; }
for rep in 1 2; do
  for n in 7 6 5 4; do
    pad=$(python3 -c "print(max(0, 163840 // ($n + 1) + 16 - $lds))")
    PPG_STEP_LDS_PAD=$pad python3 bench.py --workload c4 --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline 2>/dev/null | line $n >> $out
  done
done
cat $out
