#!/bin/bash
# On the GPU box: the walls variant's step kernel at 64 (product) / 72 / 80 registers after round 5's movement rewrite, alternating
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-8s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac']))" "$1"; }
for rep in 1 2; do
  for lib in prod walls7 walls6; do
    if [ $lib = prod ]; then unset PPG_HIP_LIB; else export PPG_HIP_LIB=$PWD/tools/_build/libppg_hip_$lib.so; fi
    python3 bench.py --workload walls --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline 2>/dev/null | line $lib
  done
done
