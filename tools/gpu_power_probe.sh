#!/bin/bash
# On the GPU box: sample power / clocks (rocm-smi) while a bench workload runs.   usage: tools/gpu_power_probe.sh TAG WORKLOAD STEPS
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; wl=${2:-policy_rollout}; steps=${3:-3000}
mkdir -p gpurun_out
out=gpurun_out/${tag}_power_$wl.txt
: > $out
( for i in $(seq 1 60); do
    echo "t=$i $(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i 'power\|sclk\|mclk\|fclk\|junction\|edge' | tr -s ' ' | tr '\n' ';' | cut -c1-600)" >> $out
    sleep 0.5
  done ) &
sampler=$!
sleep 2
python3 bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-measure-traffic > gpurun_out/${tag}_power_$wl.json 2> gpurun_out/${tag}_power_$wl.err
echo "bench done" >> $out
sleep 2
kill $sampler 2>/dev/null
wait $sampler 2>/dev/null
cat $out | cut -c1-400
tail -c 400 gpurun_out/${tag}_power_$wl.json
