#!/bin/bash
# On the GPU box: the cooperative step kernels with FEWER workgroups per CU than their 75 registers allow (6): unused LDS added to
# the launch (PPG_COOP_LDS_PAD) so that 5 / 4 / 3 fit.  18.3 KB per workgroup -> +9000 B: 5, +14000: 4 (of 160 KB), +28000: 3.
set -u
tag=${1:-r5o}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_coop_occupancy.txt
: > $out
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-10s pad %-6s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  %s' % (sys.argv[1], sys.argv[2], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], r.get('kernel')))" "$1" "$2"; }
for rep in 1 2; do
  for pad in 0 9000 14000 28000; do
    export PPG_COOP_LDS_PAD=$pad
    python3 bench.py --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic 2>/dev/null | line base $pad >> $out
    python3 bench.py --workload red_queen --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline 2>/dev/null | line red_queen $pad >> $out
  done
done
cat $out
