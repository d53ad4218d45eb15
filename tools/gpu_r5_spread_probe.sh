#!/bin/bash
# On the GPU box, as its FIRST processes: the driver's command with the observation tensors on pages spread over 32 / 128 / 256 times
# their size (bench.py --obs-spread), alternating -- does a larger stretch escape the slow placement state?
set -u
tag=${1:-r5u}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_spread_probe.txt
: > $out
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('spread %-4s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  probes %s' % (sys.argv[1], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], d['config'].get('placement_probe_us')))" "$1"; }
for rep in 1 2; do
  for n in 32 128 8 256; do
    t0=$(date +%s.%N); python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic --sustained-steps 0 --obs-spread $n 2>/dev/null | line $n >> $out; echo "   ($(python3 -c "import time; print(round(time.time() - $t0, 1))") s)" >> $out
  done
done
grep -v amdgpu $out
