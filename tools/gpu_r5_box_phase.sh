#!/bin/bash
# On the GPU box: how fast is THIS GPU at the headline step, and where does a wavefront of the cooperative kernel spend its cycles here?
# (the pool's GPUs run the same kernel at 60-82 us per 4096-env step; the bare write pattern runs the same on all of them)
set -u
tag=${1:-r5p}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_box_phase.txt
python3 bench.py --steps 300 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic > /dev/null 2>&1
python3 bench.py --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('headline on this box: %.2f M env-steps/s, %.2f us per step, kernel %.2f us, frac %.3f; probe %s; state %s' % (d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], d['config'].get('placement_probe_us'), d['config'].get('device_state_under_load')))" > $out
./tools/store_patterns4.bin 4096 36 300 0 2,4,18 1,2,9 2>&1 | grep "streams" >> $out
timeout 200 python3 tools/phase_profile.py 4096 400 --plan=4,0,2 2>&1 | grep -v amdgpu.ids >> $out
cat $out
