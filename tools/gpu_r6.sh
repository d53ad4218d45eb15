#!/bin/bash
# On the GPU box (round 6): one parameterised runner for the round's measurement steps.  usage: tools/gpu_r6.sh TAG STEP [STEP ...]
#   tests            the GPU test suite
#   ab:<workload>:<streams>:<plan>[,<plan>...]     tools/ab_plans.py, plans as name=waves.min_rows.coop (e.g. pair=2.0.0,coop42=4.0.2)
#   lib:<workload>:<steps>:<rounds>:name=lib.so[,name=lib.so...]   alternating-process A/B of library builds (tools/gpu_lib_ab.sh)
#   bench:<name>:<bench.py arguments, comma separated>      one bench.py run, its JSON line kept
#   benv:<name>:<VAR=value>:<bench.py arguments>    the same with one environment variable set for that run
#   pt:<name>:<file>:<-k expression>    selected GPU tests with their printed reports kept (pytest -s)
#   soak:<envs>:<calls>      tools/gpu_soak_coop.py: every cooperative kernel and layout, every env against its oracle on every call
#   phase:<name>:<tools/phase_profile.py arguments, comma separated>   phase shares of a wavefront's cycles (diagnostic build)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}
  case $kind in
    tests)
      python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/${tag}_pytest_gpu.log
      tail -3 gpurun_out/${tag}_pytest_gpu.log ;;
    ab)
      IFS=: read -r wl streams plans <<< "$rest"
      args=$(echo "$plans" | tr ',' ' ' | sed 's/=/:/g; s/\./,/g')
      python3 tools/ab_plans.py --workload $wl --streams $streams --rounds 5 --segment 200 $args > gpurun_out/${tag}_ab_${wl}_s${streams}.txt 2>&1
      cat gpurun_out/${tag}_ab_${wl}_s${streams}.txt | tail -8 ;;
    lib)
      IFS=: read -r wl steps rounds libs <<< "$rest"
      bash tools/gpu_lib_ab.sh $tag $rounds $wl $steps $(echo "$libs" | tr ',' ' ') | tail -12 ;;
    bench)
      IFS=: read -r name bargs <<< "$rest"
      python3 bench.py $(echo "$bargs" | tr ',' ' ') > gpurun_out/${tag}_bench_${name}.json.log 2> gpurun_out/${tag}_bench_${name}.err
      grep '^{' gpurun_out/${tag}_bench_${name}.json.log | tail -1 | cut -c1-600 ;;
    benv)
      IFS=: read -r name var bargs <<< "$rest"
      env "$var" python3 bench.py $(echo "$bargs" | tr ',' ' ') > gpurun_out/${tag}_bench_${name}.json.log 2> gpurun_out/${tag}_bench_${name}.err
      grep '^{' gpurun_out/${tag}_bench_${name}.json.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name', r.get('kernel'), 'kernel %.2f us frac %.4f value %.3f M' % (r['kernel_ms']*1e3, r['frac'], d['value']/1e6))" ;;
    pt)
      IFS=: read -r name file expr <<< "$rest"
      python3 -m pytest "$file" -m gpu -q -s -k "$expr" > gpurun_out/${tag}_pt_${name}.log 2>&1
      grep -E "^\[|passed|failed" gpurun_out/${tag}_pt_${name}.log | tail -12 ;;
    soak)
      IFS=: read -r envs calls <<< "$rest"
      python3 tools/gpu_soak_coop.py $envs $calls > gpurun_out/${tag}_soak_coop.txt 2>&1
      grep -v amdgpu.ids gpurun_out/${tag}_soak_coop.txt | tail -8 ;;
    phase)
      IFS=: read -r name pargs <<< "$rest"
      python3 tools/phase_profile.py $(echo "$pargs" | tr ',' ' ') > gpurun_out/${tag}_phase_${name}.txt 2>&1
      tail -22 gpurun_out/${tag}_phase_${name}.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
