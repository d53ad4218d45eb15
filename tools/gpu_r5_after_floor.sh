#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-14s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  %s' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], r.get('kernel')))" "$1"; }
timeout 900 python3 -m pytest tests -m gpu -x -q -k "parity or coop or fused or rollout" 2>&1 | tail -3
for rep in 1 2; do
  for s in 3 2; do
    python3 bench.py --streams $s --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic 2>/dev/null | line "base s$s"
    PPG_COOP_WGS_PER_CU=0 python3 bench.py --streams $s --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic 2>/dev/null | line "base s$s nolimit"
  done
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver"
done
