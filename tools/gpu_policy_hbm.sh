#!/bin/bash
# On the GPU box: HBM traffic of the policy kernels (FETCH_SIZE / WRITE_SIZE, separate passes; KB; FETCH x2 on gfx950 per the guide).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/ph_$c -o p -- python3 bench.py --workload policy_rollout --steps 6 --warmup 2 --no-cpu-baseline --preroll-min 128 > gpurun_out/ph_$c.log 2>&1
  python3 - "$c" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/ph_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
vals = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("ppg_policy_forward"))
half = vals[len(vals) // 2:]
print(f"{sys.argv[1]}: prey launches mean {sum(half)/len(half)/1e6:.3f} GB (raw KB counter / 1e6), predator launches mean {sum(vals[:len(vals)//2])/(len(vals)//2)/1e6:.3f} GB")
PY
  rm -rf gpurun_out/ph_$c
done
