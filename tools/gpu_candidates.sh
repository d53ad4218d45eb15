#!/bin/bash
# On the GPU box: the driver's command with K placement candidates -- wall time and what the run got.   usage: tools/gpu_candidates.sh K...
cd "$GRAFT_REPO_ROOT" || exit 1
for k in "$@"; do
  t0=$(date +%s.%N)
  python3 bench.py --steps 20 --warmup 5 --placement-candidates $k 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('K=$k', round(d['value']/1e6,2), 'M env-steps/s; kernel ms', d['roofline']['kernel_ms'], 'probes', d['config'].get('placement_candidates_us_per_step'))"
  t1=$(date +%s.%N)
  echo "   wall $(python3 -c "print(round($t1-$t0,1))") s"
done
