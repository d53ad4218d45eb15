#!/bin/bash
# On the GPU box: LDS bank-conflict cycles of the policy pipeline with one phase taken out at a time (timing-only ablation builds,
# tools/_build/libppg_hip_abl<bits>.so: 1 head, 2 staging, 4 conv1, 8 conv2, 16 conv3) -- which phase's LDS accesses conflict.
set -u
tag=${1:-r5l}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/${tag}_ablate_lds_pmc.txt
: > $out
for b in 0 1 2 4 8 16; do
  export PPG_HIP_LIB=$GRAFT_REPO_ROOT/tools/_build/libppg_hip_abl$b.so
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${tag}_p$b -o t -- python3 bench.py --workload policy_rollout --policy-open-loop --steps 12 --warmup 3 --no-cpu-baseline --no-measure-traffic > /dev/null 2> gpurun_out/${tag}_p$b.err
  rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/${tag}_q$b -o t -- python3 bench.py --workload policy_rollout --policy-open-loop --steps 12 --warmup 3 --no-cpu-baseline --no-measure-traffic > /dev/null 2>> gpurun_out/${tag}_p$b.err
  python3 - $tag $b <<'PY' >> $out
import csv, glob, sys, collections
tag, b = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}_[pq]{b}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ppg_policy" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("ablate %3s  " % b + "  ".join("%s %.4g" % (k, v) for k, v in sorted(m.items())))
PY
  rm -rf gpurun_out/${tag}_p$b gpurun_out/${tag}_q$b
done
unset PPG_HIP_LIB
cat $out
