#!/bin/bash
# On the GPU box: instruction-cache / issue counters of the policy kernels.   usage: tools/gpu_policy_pmc_misc.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-misc}
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_WAIT_ANY SQ_BUSY_CU_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc$i -o t -- python3 bench.py --workload policy_rollout --steps 12 --warmup 3 --no-cpu-baseline --no-measure-traffic > /dev/null 2> gpurun_out/${tag}_pmc$i.err || echo "set '$set' failed: $(tail -1 gpurun_out/${tag}_pmc$i.err | cut -c1-200)"
done
python3 - ${tag} <<'PY' | tee gpurun_out/${tag}_policy_pmc_misc.txt
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
PY
rm -rf gpurun_out/${tag}_pmc*
