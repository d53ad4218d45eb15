#!/bin/bash
# On the GPU box (round 6): one parameterised runner for the round's measurement steps.  usage: tools/gpu_r6.sh TAG STEP [STEP ...]
#   tests            the GPU test suite
#   ab:<workload>:<streams>:<plan>[,<plan>...]     tools/ab_plans.py, plans as name=waves.min_rows.coop (e.g. pair=2.0.0,coop42=4.0.2)
#   lib:<workload>:<steps>:<rounds>:name=lib.so[,name=lib.so...]   alternating-process A/B of library builds (tools/gpu_lib_ab.sh)
#   bench:<name>:<bench.py arguments, comma separated>      one bench.py run, its JSON line kept
#   benv:<name>:<VAR=value>:<bench.py arguments>    the same with one environment variable set for that run
#   pt:<name>:<file>:<-k expression>    selected GPU tests with their printed reports kept (pytest -s)
#   prof:<name>:<kernel>:<bench.py arguments>   rocprofv3 --kernel-trace --stats and the two PMC passes (WRITE_SIZE | FETCH_SIZE, counters only, the
#                            program directly behind `--`) of one bench.py command; mean duration / bytes per launch of <kernel>
#   sweep:<first seed>:<n>   tools/gpu_sweep.py: random configurations of every variant through the dict APIs against the oracles
#   soak:<envs>:<calls>[:walls]      tools/gpu_soak_coop.py: every cooperative kernel and layout, every env against its oracle on every call
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
    prof)
      IFS=: read -r name kern bargs <<< "$rest"
      export TMPDIR=/tmp
      A=$(echo "$bargs" | tr ',' ' ')
      rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_pt_$name -o t -- python3 bench.py $A > gpurun_out/${tag}_prof_${name}_under_trace.json.log 2> gpurun_out/${tag}_prof_${name}.err
      find gpurun_out/${tag}_pt_$name -name '*kernel_stats.csv' -exec cp {} gpurun_out/${tag}_prof_${name}_kernel_stats.csv \;
      rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pw_$name -o w -- python3 bench.py $A > /dev/null 2>> gpurun_out/${tag}_prof_${name}.err
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pf_$name -o f -- python3 bench.py $A > /dev/null 2>> gpurun_out/${tag}_prof_${name}.err
      python3 - "$tag" "$name" "$kern" > gpurun_out/${tag}_prof_${name}_summary.txt <<'PY'
import csv, glob, json, sys
tag, name, kern = sys.argv[1:4]
out = {"kernel": kern}
for f in glob.glob(f"gpurun_out/{tag}_prof_{name}_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if r["Name"] == kern:
            out["rocprofv3_calls"] = int(r["Calls"]); out["rocprofv3_mean_us"] = float(r["AverageNs"]) / 1e3
for key, pat in (("WRITE_SIZE", f"gpurun_out/{tag}_pw_{name}/**/*counter_collection.csv"), ("FETCH_SIZE", f"gpurun_out/{tag}_pf_{name}/**/*counter_collection.csv")):
    vals = [float(r["Counter_Value"]) for f in glob.glob(pat, recursive=True) for r in csv.DictReader(open(f))
            if r.get("Kernel_Name", "").split("(")[0] == kern and r.get("Counter_Name") == key]
    if vals:
        out[key + "_mean_KB_per_launch"] = sum(vals) / len(vals); out[key + "_launches"] = len(vals)
if "WRITE_SIZE_mean_KB_per_launch" in out and "FETCH_SIZE_mean_KB_per_launch" in out:
    # MI355X_MICROARCH.md, HBM section: counters in KB; FETCH_SIZE reads half of a wide coalesced read on gfx950 (doubled)
    out["hbm_bytes_per_launch_corrected"] = 1024 * (out["WRITE_SIZE_mean_KB_per_launch"] + 2 * out["FETCH_SIZE_mean_KB_per_launch"])
for l in open(f"gpurun_out/{tag}_prof_{name}_under_trace.json.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        out["bench_py_under_trace"] = {"value": d["value"], "kernel_ms": r.get("kernel_ms"), "frac": r.get("frac"), "achieved": r.get("achieved"),
                                       "counted_bytes_per_launch": r.get("counted_bytes_per_launch"), "concurrent_launches": r.get("concurrent_launches")}
print(json.dumps(out, indent=1))
PY
      rm -rf gpurun_out/${tag}_pt_$name gpurun_out/${tag}_pw_$name gpurun_out/${tag}_pf_$name
      cat gpurun_out/${tag}_prof_${name}_summary.txt ;;
    sweep)
      IFS=: read -r first n <<< "$rest"
      python3 tools/gpu_sweep.py $first $n > gpurun_out/${tag}_sweep_${first}.txt 2>&1
      grep -v amdgpu.ids gpurun_out/${tag}_sweep_${first}.txt | tail -3 ;;
    soak)
      IFS=: read -r envs calls only <<< "$rest"
      python3 tools/gpu_soak_coop.py $envs $calls $only > gpurun_out/${tag}_soak_coop.txt 2>&1
      grep -v amdgpu.ids gpurun_out/${tag}_soak_coop.txt | tail -8 ;;
    phase)
      IFS=: read -r name pargs <<< "$rest"
      python3 tools/phase_profile.py $(echo "$pargs" | tr ',' ' ') > gpurun_out/${tag}_phase_${name}.txt 2>&1
      tail -22 gpurun_out/${tag}_phase_${name}.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
