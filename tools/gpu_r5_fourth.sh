#!/bin/bash
# round 5, fourth GPU call: the pipeline's first convolution on 16x16x32 MFMAs (Conv1X): tests, bench, phase profile
set -u
tag=${1:-r5d}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_policy.py -m gpu -q 2>&1 | tail -25 > gpurun_out/${tag}_pytest_policy.log
python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy.json 2>> gpurun_out/${tag}_bench.err
PPG_POLICY_FUSED=0 python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy_fused0.json 2>> gpurun_out/${tag}_bench.err
for ip in 7000 8500 9500; do
  PPG_POLICY_ITER_P=$ip python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy_iterp$ip.json 2>> gpurun_out/${tag}_bench.err
done
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
