#!/bin/bash
# A/B of the pipeline's slot table on one box (same library: PPG_POLICY_SLOTS=0 at creation keeps the plain slot order), + LDS conflict counters
set -u
tag=${1:-r5s}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_policy.py -m gpu -q 2>&1 | tail -6 > gpurun_out/${tag}_pytest_policy.log
: > gpurun_out/${tag}_ab.txt
for rep in 1 2; do
  for sl in 1 0; do
    PPG_POLICY_SLOTS=$sl python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>> gpurun_out/${tag}_bench.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep  slot table $sl  ms_per_step %.5f  policy kernel_ms %.5f  %.2f M env-steps/s  frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'] / 1e6, d['roofline']['frac']))" >> gpurun_out/${tag}_ab.txt
  done
done
for sl in 1 0; do
  PPG_POLICY_SLOTS=$sl rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc$sl -o t -- python3 bench.py --workload policy_rollout --steps 12 --warmup 3 --no-cpu-baseline --no-measure-traffic > /dev/null 2> gpurun_out/${tag}_pmc$sl.err
  python3 - ${tag}_pmc$sl $sl <<'PY' >> gpurun_out/${tag}_ab.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ppg_policy_pipe" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c, a = (sum(acc[k]) / max(len(acc[k]), 1) for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"))
print(f"slot table {sys.argv[2]}: SQ_LDS_BANK_CONFLICT {c:.4g}  SQ_LDS_IDX_ACTIVE {a:.4g}  conflict cycles / LDS-active cycles = {c / max(a, 1):.3f}  (n={len(acc['SQ_LDS_IDX_ACTIVE'])})")
PY
  rm -rf gpurun_out/${tag}_pmc$sl
done
cat gpurun_out/${tag}_pytest_policy.log gpurun_out/${tag}_ab.txt
tail -3 gpurun_out/${tag}_bench.err
