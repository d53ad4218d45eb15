#!/bin/bash
# round 5, third GPU call: fused policy launch with the split's weights swept, the multi-wave kernels with cumulative rewards in registers
set -u
tag=${1:-r5c}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_policy.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest_policy.log
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -4 >> gpurun_out/${tag}_pytest_policy.log
PPG_POLICY_FUSED=0 python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy_fused0.json 2>> gpurun_out/${tag}_bench.err
for ip in 4500 5000 5500 6000 6500 7500; do
  PPG_POLICY_ITER_P=$ip python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy_iterp$ip.json 2>> gpurun_out/${tag}_bench.err
done
python3 bench.py --workload c4 --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline > gpurun_out/${tag}_bench_c4.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --envs 256 --steps 2000 --warmup 100 --no-cpu-baseline --sustained-steps 0 > gpurun_out/${tag}_bench_c2_256envs.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --envs 1024 --steps 2000 --warmup 100 --no-cpu-baseline --sustained-steps 0 > gpurun_out/${tag}_bench_1024envs.json 2>> gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_pytest_policy.log
for f in gpurun_out/${tag}_bench*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r.get("kernel"), r.get("kernel_ms"), r["frac"], d["config"].get("mean_agents_per_env"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -5 gpurun_out/${tag}_bench.err
