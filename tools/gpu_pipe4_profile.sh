#!/bin/bash
# On the GPU box: per-phase cycles of the four-role pipeline's wavefronts (tools/_build/libppg_hip_dprof.so = host unit with -DPPG_DIRECT_PROFILE)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
export PPG_HIP_LIB=$PWD/tools/_build/libppg_hip_dprof.so PPG_DIRECT_PROFILE_FILE=$PWD/gpurun_out/${tag}_dprof PPG_DIRECT_PROFILE_RUN=${PPG_DIRECT_PROFILE_RUN:-300}
python3 bench.py --workload policy_rollout --steps 20 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/${tag}_dprof.json 2> gpurun_out/${tag}_dprof.err
python3 - <<PY | tee gpurun_out/${tag}_pipe4_profile.txt
import numpy as np, json
names = {0: "table", 1: "conv3", 2: "barrier wait", 3: "pre", 4: "head", 5: "noise (C) / request (B1) / fetch (B2)", 11: "stage+private barrier 1", 6: "wait for B1 + park", 7: "conv1", 8: "private barrier 2",
         9: "conv2", 10: "barrier wait", 14: "logits+actions"}
print(json.loads(open("gpurun_out/${tag}_dprof.json").readlines()[-1])["roofline"])
allw = np.fromfile("gpurun_out/${tag}_dprof.fused4", dtype=np.uint64).reshape(-1, 16, 16).astype(np.float64)
for sp, a in (("prey", allw[allw[:, 0, 13] == 2]), ("pred", allw[allw[:, 0, 13] == 1])):
    a = a[a[:, 4, 15] > 0]
    its = a[:, 4, 15].mean()
    print(f"== {sp}: {len(a)} workgroups, iterations per workgroup {its:.1f}")
    for role, waves, keys in (("A", range(0, 4), (0, 3, 1, 14, 5, 2)), ("B1", range(4, 8), (0, 11, 7, 5, 8, 9, 10)), ("C", range(8, 12), (0, 3, 1, 14, 5, 2)), ("B2", range(12, 16), (0, 5, 4, 6, 10))):
        tot = a[:, waves, :13].sum(axis=2) + a[:, waves, 14]
        print(f"  role {role}: cycles per wavefront {tot.mean():.0f}; per iteration {tot.mean() / its:.0f}: " +
              "  ".join(f"{names[k]} {a[:, waves, k].mean() / its:.0f}" for k in keys))
PY
rm -f gpurun_out/${tag}_dprof.fused4 gpurun_out/${tag}_dprof.fused
