#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('pad %-5s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  %s' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], r.get('kernel')))" "$1"; }
for rep in 1 2 3; do
  for pad in 0 208; do
    PPG_STEP_LDS_PAD=$pad python3 bench.py --workload c4 --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline 2>/dev/null | line $pad
  done
done
