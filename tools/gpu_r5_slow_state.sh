#!/bin/bash
# On the GPU box, first processes: the driver's command with spread 32 (default) / 0 (torch's allocator) / 8 / 128 and with six workgroups
# per CU -- what helps when the box is in its slow state (probes >= 79 us)?
set -u
tag=${1:-r5ss}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_slow_state.txt
: > $out
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-14s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  probes %s' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], d['config'].get('placement_probe_us')))" "$1"; }
for rep in 1 2; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic --sustained-steps 0 2>/dev/null | line "spread32" >> $out
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic --sustained-steps 0 --obs-spread 0 2>/dev/null | line "spread0" >> $out
  PPG_COOP_WGS_PER_CU=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic --sustained-steps 0 2>/dev/null | line "spread32_6wg" >> $out
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic --sustained-steps 0 --obs-spread 8 2>/dev/null | line "spread8" >> $out
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic --sustained-steps 0 --streams 2 2>/dev/null | line "spread32_2sub" >> $out
done
cat $out
