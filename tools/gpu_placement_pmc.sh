#!/bin/bash
# On the GPU box: do env groups in differently placed buffers differ in address-translation misses?  rocprofv3 --pmc over tools/exp_placement.py
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-pl}
shift
for ctr in "$@"; do
  rm -rf /tmp/plpmc
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/plpmc -o p -- python3 tools/exp_placement.py --groups 5 --rounds 2 --segment 60 --preroll 512 > gpurun_out/${tag}_pmc_$ctr.log 2>&1
  python3 - "$ctr" gpurun_out/${tag}_pmc_$ctr.log <<'PY' | tee -a gpurun_out/${tag}_placement_pmc.txt
import csv, glob, sys, collections
fs = glob.glob("/tmp/plpmc/**/*counter_collection.csv", recursive=True)
log = open(sys.argv[2]).read()
print("#####", sys.argv[1])
print("\n".join(l for l in log.splitlines() if l.startswith("group") and "fill" not in l))
if not fs:
    print("no counter output", log[-300:]); raise SystemExit
rows = [r for r in csv.DictReader(open(fs[0])) if r["Kernel_Name"].startswith("ppgc_step")]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
vals = [float(r["Counter_Value"]) for r in rows]
# dispatch order: 5 groups x 512 pre-roll steps x 3 sub-batches, then rounds: (20 + 60) steps per group, forward then backward
per = 3
pre = 5 * 512 * per
seg = 80 * per
order = list(range(5)) + list(reversed(range(5)))
acc = collections.defaultdict(list)
for i, g in enumerate(order):
    chunk = vals[pre + i * seg + 20 * per: pre + (i + 1) * seg]
    if chunk:
        acc[g].append(sum(chunk) / len(chunk))
for g in sorted(acc):
    print(f"group {g}: mean {sys.argv[1]} per dispatch {[round(v, 1) for v in acc[g]]}")
PY
done
