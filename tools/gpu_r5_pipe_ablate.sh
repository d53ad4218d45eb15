#!/bin/bash
# On the GPU box: timing-only ablations of the two-role pipeline (libraries built on the CPU: for b in 0 1 2 4 8 16 32 64 128 255;
#   python3 -c "import __graft_entry__ as g; g.build_hip_variant('tools/_build/libppg_hip_abl$b.so', ['-DPPG_PIPE_ABLATE=$b'])")
# What the step costs with one phase taken out says how much of the iteration that phase is RESPONSIBLE for -- the two roles share a
# SIMD, so a phase's own cycle count (tools/gpu_pipe_profile.sh) is not that.
set -u
tag=${1:-r5abl}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/${tag}_ablate.txt
for b in 0 1 2 4 8 16 32 64 128 255 0; do
  PPG_HIP_LIB=$GRAFT_REPO_ROOT/tools/_build/libppg_hip_abl$b.so python3 bench.py --workload policy_rollout --policy-open-loop --steps 60 --warmup 10 --no-cpu-baseline 2>> gpurun_out/${tag}_bench.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ablate %3d  ms_per_step %.5f  policy kernel_ms %.5f  agents per env %.2f' % ($b, d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['mean_agents_per_env']))" >> gpurun_out/${tag}_ablate.txt
done
cat gpurun_out/${tag}_ablate.txt
