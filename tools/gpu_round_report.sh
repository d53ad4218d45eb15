#!/bin/bash
# On the GPU box: the end-of-round evidence run.  usage: tools/gpu_round_report.sh <tag>   (ONE run per round; the experiments of a round go through tools/gpu_r6.sh)
set -u
tag=${1:-r06}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
# the driver's command first, on the box as it comes (pytest afterwards: a box is 15-20 % slower once it has been under load)
bash tools/gpu_profile_driver_cmd.sh ${tag} > gpurun_out/${tag}_profile.log 2>&1
./tools/store_patterns4.bin 4096 36 300 0 2,4,18 1,2,9 2>&1 | grep "streams" > gpurun_out/${tag}_bare_pattern.txt
python3 tools/ab_plans.py --streams 3 --rounds 5 coop42:4,0,2 pair:2,0,0 one:1,0,0 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_ab_plans.txt
python3 tools/ab_plans.py --workload c4 --streams 3 --rounds 5 auto:0,0,0 pair:2,0,0 2>&1 | grep -v amdgpu.ids >> gpurun_out/${tag}_ab_plans.txt
python3 tools/ab_plans.py --workload red_queen --streams 3 --rounds 5 coop42:4,0,2 w2:4,72,0 2>&1 | grep -v amdgpu.ids >> gpurun_out/${tag}_ab_plans.txt
./tools/store_patterns4.bin 4096 36 300 0 2,4,18 1,2,9 2>&1 | grep "streams" >> gpurun_out/${tag}_bare_pattern.txt
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/${tag}_pytest.log
for w in c4 red_queen drive walls; do
  python3 bench.py --workload $w --steps 1000 --warmup 100 --sustained-steps 0 > gpurun_out/${tag}_bench_$w.json 2>> gpurun_out/${tag}_bench.err
done
python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --cpu-seconds 8 > gpurun_out/${tag}_bench_policy_rollout.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --workload policy_rollout --obs-dtype f64 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy_rollout_f64.json 2>> gpurun_out/${tag}_bench.err
for a in fc256 r3 depth; do   # the same encoder with head_fcnet_hiddens [256, 256]; rounds 2-3's network; (R - 1) // 2 convolutions
  python3 bench.py --workload policy_rollout --policy-arch $a --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_policy_rollout_$a.json 2>> gpurun_out/${tag}_bench.err
done
python3 bench.py --envs 256 --steps 2000 --warmup 100 --no-cpu-baseline --sustained-steps 0 > gpurun_out/${tag}_bench_c2_256envs.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --workload dict_api > gpurun_out/${tag}_bench_dict_api.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --steps 1000 --warmup 100 --sustained-steps 0 --no-cpu-baseline --no-measure-traffic > gpurun_out/${tag}_bench_headline_1000_steps.json 2>> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_ptrace -o t -- python3 bench.py --workload policy_rollout --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_policy_under_trace.json 2> gpurun_out/${tag}_ptrace.err
find gpurun_out/${tag}_ptrace -name '*kernel_stats.csv' -exec cp {} gpurun_out/${tag}_kernel_stats_policy.csv \;
rm -rf gpurun_out/${tag}_ptrace
# MFMA-busy share of the policy kernels (counters in runs of their own, two at a time) and package power / clocks during a policy run
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${tag}_ppmc$i -o t -- python3 bench.py --workload policy_rollout --steps 12 --warmup 3 --no-cpu-baseline --no-measure-traffic > /dev/null 2> gpurun_out/${tag}_ppmc$i.err
done
python3 - ${tag} <<'PY' > gpurun_out/${tag}_policy_pmc.txt 2>&1
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_ppmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "ppg_policy" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k, "  ".join(f"{c} {sum(v) / len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CU_CYCLES" in cs:
        m, b = (sum(cs[c]) / len(cs[c]) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"))
        print(f"   MFMA busy cycles / (4 x CU busy cycles) = {m / (4 * b):.3f}   (MFMA_BUSY is summed over the SIMDs: MI355X_MICROARCH.md)")
    if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_LDS_IDX_ACTIVE" in cs:
        c, a = (sum(cs[k_]) / len(cs[k_]) for k_ in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"))
        print(f"   LDS bank-conflict cycles / LDS-active cycles = {c / a:.3f}")
PY
rm -rf gpurun_out/${tag}_ppmc*
bash tools/gpu_power_probe.sh ${tag} policy_rollout 3000 > /dev/null 2>&1
cat gpurun_out/${tag}_policy_pmc.txt
cat gpurun_out/${tag}_pytest.log
for f in gpurun_out/${tag}_bench_*.json gpurun_out/${tag}_bench.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], r.get("kernel"), r["kernel_ms"], r["frac"], r.get("frac_sustained"), d["config"].get("mean_agents_per_env"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -c 900 gpurun_out/${tag}_summary.log
cat gpurun_out/${tag}_ab_plans.txt gpurun_out/${tag}_bare_pattern.txt
PPG_DIRECT_PROFILE_RUN=300 bash tools/gpu_pipe_profile.sh ${tag} > /dev/null 2>&1
cat gpurun_out/${tag}_pipe_profile.txt | grep "==\|per iteration"
