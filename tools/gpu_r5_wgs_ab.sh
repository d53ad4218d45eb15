#!/bin/bash
# On the GPU box: five (default) vs six (PPG_COOP_WGS_PER_CU=0) vs four cooperative workgroups per CU, alternating processes, headline workload
set -u
tag=${1:-r5q}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_wgs_ab.txt
: > $out
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-6s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  probe %s' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], d['config'].get('placement_probe_us', {}).get('min')))" "$1"; }
python3 bench.py --steps 300 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic > /dev/null 2>&1
for rep in 1 2 3 4; do
  for n in 5 0 4; do
    PPG_COOP_WGS_PER_CU=$n python3 bench.py --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic 2>/dev/null | line "wgs$n" >> $out
  done
done
cat $out
