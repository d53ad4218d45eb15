#!/bin/bash
# On the GPU box: A/B of experimental builds of libppg_hip.so on the policy rollout (alternating processes, same box).
#   usage: tools/gpu_policy_ab.sh TAG ROUNDS name=path/to/lib.so [name=...]     (name "base" = the product library)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out
out=gpurun_out/${tag}_policy_ab.txt
: > $out
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ "$name" = base ]; then unset PPG_HIP_LIB; else export PPG_HIP_LIB=$PWD/$lib; fi
    python3 bench.py --workload policy_rollout --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_pab.json 2> gpurun_out/${tag}_pab.err
    python3 - "$name" "$r" gpurun_out/${tag}_pab.json >> $out <<'PY'
import json, sys
for l in open(sys.argv[3]):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(f"{sys.argv[1]:12s} round {sys.argv[2]}  policy kernels {r['kernel_ms']*1e3:8.1f} us  frac {r['frac']:.4f}  step {d['ms_per_step']*1e3:8.1f} us  value {d['value']/1e6:.3f} M")
        break
else:
    print(sys.argv[1], "no JSON:", open(sys.argv[3].replace('.json', '.err')).read()[-300:])
PY
  done
done
cat $out
