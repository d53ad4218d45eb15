#!/bin/bash
# round 5, first GPU call: the suite, the dict-API leg (before / after data movement), baselines of the workloads this round works on
set -u
tag=${1:-r5a}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --workload dict_api > gpurun_out/${tag}_bench_dict_api.json 2>> gpurun_out/${tag}_bench.err
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${tag}_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/${tag}_pytest.log
python3 bench.py --workload c4 --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline > gpurun_out/${tag}_bench_c4.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --envs 256 --steps 2000 --warmup 100 --no-cpu-baseline --sustained-steps 0 > gpurun_out/${tag}_bench_c2_256envs.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy_rollout.json 2>> gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_pytest.log
for f in gpurun_out/${tag}_bench*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r.get("kernel"), r.get("kernel_ms"), r["frac"], d["config"].get("mean_agents_per_env"), d["config"].get("single_env"), d["config"].get("single_env_rounds_1_to_4_data_movement"), d["config"].get("vector_env"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -5 gpurun_out/${tag}_bench.err
