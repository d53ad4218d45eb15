#!/bin/bash
# On the GPU box: SQ counters of the policy kernels (bench.py --workload policy_rollout, short run).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d gpurun_out/pol_pmc -o p -- python3 bench.py --workload policy_rollout --steps 6 --warmup 2 --no-cpu-baseline --preroll-min 128 > gpurun_out/pol_pmc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pol_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("ppg_policy_forward"):
        acc[(r["Kernel_Name"], r["Grid_Size"] if "Grid_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    # predators and prey alternate: split by dispatch parity is not available here, print means of the larger half (prey)
    for name, vals in sorted(v.items()):
        vals = sorted(vals)
        half = vals[len(vals) // 2:]
        print(f"  {name:28s} mean(all) {sum(vals)/len(vals):.4g}   mean(upper half = prey launches) {sum(half)/len(half):.4g}")
PY
rm -rf gpurun_out/pol_pmc
