#!/bin/bash
# On the GPU box: per-tile phase stamps of one policy step (needs predpreygrass_amd/csrc/libppg_hip_exp.so = a -DPPG_EXPERIMENTS build)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-tl}
mkdir -p gpurun_out
export PPG_HIP_LIB=$PWD/predpreygrass_amd/csrc/libppg_hip_exp.so PPG_POLICY_TIMELINE=$PWD/gpurun_out/${tag}_timeline PPG_POLICY_TIMELINE_RUN=${2:-450}
python3 bench.py --workload policy_rollout --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_timeline_bench.json 2> gpurun_out/${tag}_timeline.err
tail -c 600 gpurun_out/${tag}_timeline_bench.json
python3 tools/policy_timeline.py gpurun_out/${tag}_timeline | tee gpurun_out/${tag}_timeline.txt
