#!/bin/bash
# On the GPU box: where the time of a policy_rollout step goes BETWEEN kernels (rocprofv3 --kernel-trace timestamps).
# usage: tools/gpu_policy_gaps.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-gaps}
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_trace -o t -- python3 bench.py --workload policy_rollout --steps 60 --warmup 10 --no-cpu-baseline --no-measure-traffic > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_trace.err
python3 - ${tag} <<'PY' | tee gpurun_out/${tag}_policy_gaps.txt
import csv, glob, sys
tag = sys.argv[1]
rows = []
for f in glob.glob(f"gpurun_out/{tag}_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# the last 40 steps: a step = plan2, pipe16, pipe8, then the step kernels
idx = [i for i, r in enumerate(rows) if r[2].startswith("ppg_policy_plan")]
idx = idx[-41:]
tot = {"plan": 0, "gap plan->prey": 0, "prey": 0, "gap prey->pred": 0, "pred": 0, "gap pred->first step kernel": 0, "step kernels (first start to last end)": 0, "gap last step kernel->next plan": 0}
n = 0
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    names = [s[2] for s in seg]
    if len(seg) < 4 or not names[1].startswith("ppg_policy_pipe") or not names[2].startswith("ppg_policy_pipe"):
        continue
    plan, prey, pred = seg[0], seg[1], seg[2]
    steps = [s for s in seg[3:] if "step" in s[2]]
    if not steps:
        continue
    n += 1
    tot["plan"] += plan[1] - plan[0]
    tot["gap plan->prey"] += prey[0] - plan[1]
    tot["prey"] += prey[1] - prey[0]
    tot["gap prey->pred"] += pred[0] - prey[1]
    tot["pred"] += pred[1] - pred[0]
    tot["gap pred->first step kernel"] += min(s[0] for s in steps) - pred[1]
    tot["step kernels (first start to last end)"] += max(s[1] for s in steps) - min(s[0] for s in steps)
    tot["gap last step kernel->next plan"] += rows[b][0] - max(s[1] for s in steps)
print(f"{n} steps under the kernel trace; microseconds per step:")
for k, v in tot.items():
    print(f"  {k:45s} {v / n / 1000:8.2f}")
print(f"  {'sum':45s} {sum(tot.values()) / n / 1000:8.2f}")
PY
tail -c 300 gpurun_out/${tag}_bench.json
rm -rf gpurun_out/${tag}_trace
