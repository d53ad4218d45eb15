#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of the other workloads of bench.py (1000 timed steps each), one csv per workload.
# usage: tools/gpu_variant_stats.sh <tag>
set -u
tag=${1:-r02v}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in c4 red_queen drive walls; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_vt_$w -o t -- python3 bench.py --workload $w --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/${tag}_under_trace_$w.json 2> /dev/null
  find gpurun_out/${tag}_vt_$w -name '*kernel_stats.csv' -exec cp {} gpurun_out/${tag}_kernel_stats_$w.csv \;
  rm -rf gpurun_out/${tag}_vt_$w
  head -3 gpurun_out/${tag}_kernel_stats_$w.csv | cut -c1-140
  grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*' gpurun_out/${tag}_under_trace_$w.json | head -2 | tr '\n' ' '; echo
done
