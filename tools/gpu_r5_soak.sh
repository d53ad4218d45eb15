#!/bin/bash
# On the GPU box: randomised differential soak of the round-5 library (movement phase rewritten): tools/gpu_sweep.py in parallel
# processes (random configurations of the three families through the dict APIs vs the oracles, call by call) + tools/gpu_long_rollouts.py
set -u
tag=${1:-r5s}; procs=${2:-48}; seeds=${3:-300}; first=${4:-500000}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/${tag}_soak
timeout ${7:-420} python3 -u tools/gpu_long_rollouts.py ${5:-64} ${6:-300} > gpurun_out/${tag}_soak/long_rollouts.txt 2>&1 &
for i in $(seq 0 $((procs - 1))); do
  timeout ${7:-420} python3 tools/gpu_sweep.py $((first + i * seeds)) $seeds > gpurun_out/${tag}_soak/sweep_$i.txt 2>&1 &
done
wait
for f in gpurun_out/${tag}_soak/sweep_*.txt; do grep "^ok\|^progress" $f | tail -1; done | sed 's/^progress/ok/; s/fails \([0-9][0-9]*\) /fails [] /' | python3 -c "
import sys, re, ast
tot = {'base': 0, 'gen2': 0, 'walls': 0}; fails = []
for l in sys.stdin:
    m = re.match(r\"ok (\{.*?\}) fails (\[.*\]) \", l)
    c = ast.literal_eval(m.group(1)); f = ast.literal_eval(m.group(2))
    for k in tot: tot[k] += c[k]
    fails += f
print('soak: configurations passed', tot, 'failures', len(fails), fails[:5])" > gpurun_out/${tag}_soak_summary.txt
echo "processes that were cut off before their last seed: $(grep -L "^ok" gpurun_out/${tag}_soak/sweep_*.txt | wc -l) of $procs; tracebacks: $(cat gpurun_out/${tag}_soak/sweep_*.txt | grep -c Traceback)" >> gpurun_out/${tag}_soak_summary.txt
cat gpurun_out/${tag}_soak/long_rollouts.txt | grep -v amdgpu >> gpurun_out/${tag}_soak_summary.txt
cat gpurun_out/${tag}_soak_summary.txt
