#!/bin/bash
# On the GPU box: PPG_COOP_LDS_PAD 0 / 9000 / 14000 (6 / 5 / 4 workgroups per CU), alternating, headline workload; then the driver's command
set -u
tag=${1:-r5o2}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_coop_occupancy.txt
: > $out
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-10s pad %-6s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  %s' % (sys.argv[1], sys.argv[2], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], r.get('kernel')))" "$1" "$2"; }
python3 bench.py --steps 300 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic > /dev/null 2>&1   # (the first process on a box runs slower)
for rep in 1 2 3 4; do
  for pad in 0 9000 14000; do
    export PPG_COOP_LDS_PAD=$pad
    python3 bench.py --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic 2>/dev/null | line base $pad >> $out
  done
done
for rep in 1 2 3; do
  for pad in 0 14000; do
    export PPG_COOP_LDS_PAD=$pad
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic 2>/dev/null | line driver $pad >> $out
  done
done
cat $out
