#!/bin/bash
# On the GPU box: LDS counters of the policy kernels (one rocprofv3 --pmc pass per pair of counters).   usage: tools/gpu_policy_lds_pmc.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-lds}
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc$i -o t -- python3 bench.py --workload policy_rollout --steps 12 --warmup 3 --no-cpu-baseline --no-measure-traffic > /dev/null 2> gpurun_out/${tag}_pmc$i.err
done
python3 - ${tag} <<'PY' | tee gpurun_out/${tag}_policy_lds_pmc.txt
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "ppg_policy_pipe" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(k, "  ".join(f"{c} {v:.4g}" for c, v in sorted(m.items())))
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
        print(f"   LDS bank-conflict cycles / LDS active cycles = {m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
    if "SQ_ACTIVE_INST_LDS" in m and "SQ_BUSY_CU_CYCLES" in m:
        print(f"   LDS instruction-active cycles x 4 / CU busy cycles = {4 * m['SQ_ACTIVE_INST_LDS'] / m['SQ_BUSY_CU_CYCLES']:.3f}   (ACTIVE_INST counts quad-cycles: MI355X_MICROARCH.md)")
    if "SQ_WAIT_INST_LDS" in m and "SQ_WAVE_CYCLES" in m:
        print(f"   wave-cycles waiting for LDS / wave-cycles = {m['SQ_WAIT_INST_LDS'] / m['SQ_WAVE_CYCLES']:.3f};  waiting for anything = {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}")
PY
rm -rf gpurun_out/${tag}_pmc*
