#!/bin/bash
# On the GPU box: same-box A/B of the movement phase (counted touches, wishes computed once) against the library before it.
# usage: tools/gpu_r5_move_ab.sh <tag>     (tools/_build/libppg_hip_oldmove.so = the step kernels of the commit before)
set -u
tag=${1:-r5m}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_move_ab.txt
: > $out
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-10s %-4s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  %s' % (sys.argv[1], sys.argv[2], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], r.get('kernel')))" "$1" "$2"; }
for rep in 1 2; do
  for w in walls red_queen c4 drive; do
    for lib in old new; do
      if [ $lib = old ]; then export PPG_HIP_LIB=$PWD/tools/_build/libppg_hip_oldmove.so; else unset PPG_HIP_LIB; fi
      python3 bench.py --workload $w --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline 2>/dev/null | line $w $lib >> $out
    done
  done
  for lib in old new; do
    if [ $lib = old ]; then export PPG_HIP_LIB=$PWD/tools/_build/libppg_hip_oldmove.so; else unset PPG_HIP_LIB; fi
    python3 bench.py --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic 2>/dev/null | line base $lib >> $out
    python3 bench.py --envs 256 --steps 2000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic 2>/dev/null | line c2_256 $lib >> $out
    python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | line policy $lib >> $out
  done
done
unset PPG_HIP_LIB
cat $out
