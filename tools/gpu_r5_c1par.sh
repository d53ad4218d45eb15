#!/bin/bash
# On the GPU box: conv1's tiles by cell parity (tools/_build/libppg_hip_c1par.so: -DPPG_PIPE_CONV1_PARITY=1) against the product library:
# policy tests, timing A/B (alternating), LDS conflict counters
set -u
tag=${1:-r5k}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/${tag}_c1par.txt
PPG_HIP_LIB=$PWD/tools/_build/libppg_hip_c1par.so timeout 400 python3 -m pytest tests/test_policy.py -m gpu -x -q 2>&1 | tail -2 > $out
bash tools/gpu_r5_ab_libs.sh ${tag} base c1par base c1par >> $out 2>&1
for lib in base c1par; do
  if [ $lib = base ]; then unset PPG_HIP_LIB; else export PPG_HIP_LIB=$PWD/tools/_build/libppg_hip_$lib.so; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${tag}_p$lib -o t -- python3 bench.py --workload policy_rollout --steps 12 --warmup 3 --no-cpu-baseline --no-measure-traffic > /dev/null 2> gpurun_out/${tag}_p$lib.err
  python3 - $tag $lib <<'PY' >> $out
import csv, glob, sys, collections
tag, b = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}_p{b}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ppg_policy" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("%-6s " % b + "  ".join("%s %.4g" % (k, v) for k, v in sorted(m.items())), " ratio %.3f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
  rm -rf gpurun_out/${tag}_p$lib
done
unset PPG_HIP_LIB
grep -v amdgpu $out
