#!/bin/bash
# first GPU run of the cooperative kernels: parity tests, then an A/B of plans and sub-batch counts on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "cooperative or default_wave_plans or multiwave_step" > gpurun_out/r3a_pytest_coop.log 2>&1
tail -5 gpurun_out/r3a_pytest_coop.log
B="python3 bench.py --no-cpu-baseline --steps 1500 --warmup 200"
for rep in 1 2; do
for cfg in "coop443:--streams 3" "coop442:--streams 2" "coop441:--streams 1" "coop444:--streams 4" "pair3:--streams 3 --wave-plan 2,0,0" "one3:--streams 3 --wave-plan 1,0,0" "coop84_3:--streams 3 --wave-plan 8,0,4" "coop42_3:--streams 3 --wave-plan 4,0,2" "coop88_3:--streams 3 --wave-plan 8,0,8"; do
  name=${cfg%%:*}; args=${cfg#*:}
  $B $args > gpurun_out/r3a_${name}_$rep.json 2>> gpurun_out/r3a_err.log
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r3a_${name}_$rep.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("${name} rep $rep: %.1f M env-steps/s, kernel %s %.1f us, frac %.3f, rows/env %.2f" % (d["value"]/1e6, r["kernel"], r["kernel_ms"]*1e3, r["frac"], d["config"]["mean_agents_per_env"]))
PY
done
done
