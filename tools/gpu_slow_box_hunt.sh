#!/bin/bash
# On the GPU box: if this is one of the pool's SLOW GPUs (every placement of the headline workload probes above THRESH us per step), compare
# what else could be chosen there: sub-batch counts, wave plans, allocation spreads.  On a normal GPU: say so and stop (cheap).
# usage: tools/gpu_slow_box_hunt.sh [THRESH=84]
cd "$GRAFT_REPO_ROOT" || exit 1
thresh=${1:-84}
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic --sustained-steps 0 --fused-steps 0 2>/dev/null | tail -1 > gpurun_out/hunt_bench.json
probe=$(python3 -c "import json; d=json.loads(open('gpurun_out/hunt_bench.json').read()); print(d['config']['placement_probe_us']['min'], round(d['value']/1e6,2), d['roofline']['kernel_ms'])")
echo "HUNT probe_min value kernel_ms: $probe"
slow=$(python3 -c "print(1 if float('$probe'.split()[0]) >= $thresh else 0)")
if [ "$slow" != "1" ]; then echo "HUNT normal GPU"; exit 0; fi
echo "HUNT SLOW GPU"
for st in 1 2 3 4; do
  python3 tools/ab_plans.py --streams $st --rounds 3 coop42:4,0,2 pair:2,0,0 coop44:4,0,4 one:1,0,0 coop41:4,0,1 2>&1 | grep -v amdgpu.ids
done
for sp in 0 4 32 128; do
  python3 tools/ab_plans.py --streams 3 --rounds 3 --obs-spread $sp coop42:4,0,2 2>&1 | grep -v amdgpu.ids | sed "s/^/spread $sp: /"
done
./tools/store_patterns4.bin 4096 36 300 0 2,4,18 1,2,9 2>&1 | grep "streams"
