#!/bin/bash
# On the GPU box: policy_rollout with several builds of the library, interleaved on the same GPU (the boxes differ by 5-10 %).
# usage: tools/gpu_ab_policy.sh name=path/to/lib.so ...     (paths relative to the repo root)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { PPG_HIP_LIB=$PWD/$1 timeout 120 python3 bench.py --workload policy_rollout --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for i in 1 2 3; do for a in "$@"; do run "${a#*=}" "${a%%=*}"; done; done
