#!/bin/bash
# round 5, second GPU call: the fused policy launch (tests + bench, A/B against the separate launches), BASELINE config 4's bare write
# patterns, phase profiles of config 2 (256 envs) and config 4, the dict-API leg with the lighter Python
set -u
tag=${1:-r5b}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_policy.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest_policy.log
for f in 1 0; do
  PPG_POLICY_FUSED=$f python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy_fused$f.json 2>> gpurun_out/${tag}_bench.err
done
for ip in 5000 6000 7000 8000; do
  PPG_POLICY_ITER_P=$ip python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy_iterp$ip.json 2>> gpurun_out/${tag}_bench.err
done
./tools/store_patterns_c4.bin 4096 39 300 15 > gpurun_out/${tag}_store_patterns_c4.txt 2>&1
python3 tools/phase_profile.py 256 600 > gpurun_out/${tag}_phase_c2_256envs.txt 2>&1
python3 tools/phase_profile.py 4096 600 --c4 > gpurun_out/${tag}_phase_c4.txt 2>&1
python3 bench.py --workload dict_api > gpurun_out/${tag}_bench_dict_api.json 2>> gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_pytest_policy.log
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
cat gpurun_out/${tag}_store_patterns_c4.txt
grep -v amdgpu.ids gpurun_out/${tag}_phase_c2_256envs.txt
grep -v amdgpu.ids gpurun_out/${tag}_phase_c4.txt
tail -5 gpurun_out/${tag}_bench.err
