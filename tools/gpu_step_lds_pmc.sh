#!/bin/bash
# On the GPU box: LDS / wait counters of the step kernels of a bench workload (one rocprofv3 --pmc pass per counter pair).
# usage: tools/gpu_step_lds_pmc.sh TAG [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-slds}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc$i -o t -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-measure-traffic --sustained-steps 0 --fused-steps 0 --placement-candidates 1 "$@" > /dev/null 2> gpurun_out/${tag}_pmc$i.err
done
python3 - ${tag} <<'PY' | tee gpurun_out/${tag}_step_lds_pmc.txt
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "_step_q" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(k, "  ".join(f"{c} {v:.4g}" for c, v in sorted(m.items())))
    g = lambda a: m.get(a, 0.0)
    print(f"   LDS bank-conflict cycles / LDS active cycles = {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f};  LDS active cycles / CU busy cycles = {g('SQ_LDS_IDX_ACTIVE') / max(g('SQ_BUSY_CU_CYCLES'), 1):.3f}")
    print(f"   wave-cycles waiting for LDS / wave-cycles = {g('SQ_WAIT_INST_LDS') / max(g('SQ_WAVE_CYCLES'), 1):.3f};  waiting for anything = {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f};  VALU-active (x4) / CU busy = {4 * g('SQ_ACTIVE_INST_VALU') / max(g('SQ_BUSY_CU_CYCLES'), 1):.3f}")
PY
rm -rf gpurun_out/${tag}_pmc*
