#!/bin/bash
# On the GPU box: the cooperative step kernels capped at 72 / 64 registers (7 / 8 wavefronts per SIMD: 14 / 16 envs per CU resident
# instead of 12), same-box A/B against the product library.  tools/_build/libppg_hip_coop{7,8}.so
set -u
tag=${1:-r5c}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_coopcap_ab.txt
: > $out
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-10s %-6s %8.2f M env-steps/s  %7.2f us per step  kernel %7.2f us  frac %.3f  %s' % (sys.argv[1], sys.argv[2], d['value'] / 1e6, d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3, r['frac'], r.get('kernel')))" "$1" "$2"; }
for rep in 1 2; do
  for lib in prod coop7 coop8; do
    if [ $lib = prod ]; then unset PPG_HIP_LIB; else export PPG_HIP_LIB=$PWD/tools/_build/libppg_hip_$lib.so; fi
    python3 bench.py --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic 2>/dev/null | line base $lib >> $out
    python3 bench.py --workload red_queen --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline 2>/dev/null | line red_queen $lib >> $out
    python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | line policy $lib >> $out
  done
done
unset PPG_HIP_LIB
cat $out
