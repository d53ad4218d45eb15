#!/bin/bash
# On the GPU box: per-wave cycle profile of a policy tile with parts of the convolution loops switched off (results meaningless,
# timing only; needs libppg_hip_exp.so = -DPPG_EXPERIMENTS).   usage: tools/gpu_policy_ablate.sh TAG SKIP [SKIP ...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
export PPG_POLICY_TIMELINE_RUN=300
for skip in "$@"; do   # (one library per ablation: predpreygrass_amd/csrc/libppg_hip_abl<bits>.so = -DPPG_EXPERIMENTS -DPPG_ABLATE=<bits>)
  export PPG_HIP_LIB=$PWD/predpreygrass_amd/csrc/libppg_hip_abl$skip.so PPG_POLICY_TIMELINE=$PWD/gpurun_out/${tag}_abl_$skip
  python3 bench.py --workload policy_rollout --steps 20 --warmup 10 --no-cpu-baseline --preroll-min 300 > gpurun_out/${tag}_abl_$skip.json 2> gpurun_out/${tag}_abl_$skip.err
  echo "######## PPG_ABLATE=$skip" | tee -a gpurun_out/${tag}_ablate.txt
  python3 tools/policy_timeline.py gpurun_out/${tag}_abl_$skip | grep -A16 "^== prey" | grep -v "workgroups:\|tiles in flight\|round \|small tiles" | tee -a gpurun_out/${tag}_ablate.txt
  rm -f gpurun_out/${tag}_abl_$skip.prey gpurun_out/${tag}_abl_$skip.pred
done
