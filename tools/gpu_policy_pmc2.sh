#!/bin/bash
# On the GPU box: where do the policy kernel's waves wait?  Two counter passes over a short policy_rollout run.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pp_$name -o p -- python3 bench.py --workload policy_rollout --steps 6 --warmup 2 --no-cpu-baseline --preroll-min 128 > gpurun_out/pp_$name.log 2>&1
  python3 - "$name" <<'PY'
import csv, glob, collections, sys
fs = glob.glob(f"gpurun_out/pp_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)
if not fs:
    print(sys.argv[1], "no output:", open(f"gpurun_out/pp_{sys.argv[1]}.log").read()[-400:]); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if r["Kernel_Name"].startswith("ppg_policy_forward"):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, vals in sorted(acc.items()):
    vals = sorted(vals); half = vals[len(vals) // 2:]
    print(f"  {name:28s} prey launches mean {sum(half)/len(half):.4g}")
PY
  rm -rf gpurun_out/pp_$name
}
run a SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run b SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU
run c SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
