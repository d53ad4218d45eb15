#!/bin/bash
# On the GPU box: A/B of two builds of libppg_hip.so on one bench workload (alternating processes, same box).
#   usage: tools/gpu_lib_ab.sh TAG ROUNDS WORKLOAD STEPS name=path/to/lib.so[@waves.min_rows.coop_envs] [name=...]
#   (a lib of "base" = the product library; @...: bench.py --wave-plan for that leg)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; rounds=$2; wl=$3; steps=$4; shift 4
mkdir -p gpurun_out
out=gpurun_out/${tag}_lib_ab_$wl.txt
: > $out
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}; plan=
    case $lib in *@*) plan="--wave-plan $(echo ${lib#*@} | tr . ,)"; lib=${lib%%@*};; esac
    if [ "$name" = base ] || [ "$lib" = base ]; then unset PPG_HIP_LIB; else export PPG_HIP_LIB=$PWD/$lib; fi
    python3 bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-measure-traffic --sustained-steps 0 --fused-steps 0 $plan > gpurun_out/${tag}_lab.json 2> gpurun_out/${tag}_lab.err
    python3 - "$name" "$r" gpurun_out/${tag}_lab.json >> $out <<'PY'
import json, sys
for l in open(sys.argv[3]):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(f"{sys.argv[1]:12s} round {sys.argv[2]}  {r.get('kernel')}  kernel {r['kernel_ms']*1e3:8.2f} us  frac {r['frac']:.4f}  step {d['ms_per_step']*1e3:8.2f} us  value {d['value']/1e6:.3f} M")
        break
else:
    print(sys.argv[1], "no JSON:", open(sys.argv[3].replace('.json', '.err')).read()[-300:])
PY
  done
done
cat $out
