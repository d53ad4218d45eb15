#!/bin/bash
# On the GPU box: where the wavefronts of the two-role pipeline kernels (ppg_policy_pipe.h) spend their cycles (diagnostic library built
# on the CPU:  python -c "import os, __graft_entry__ as g; g.build_hip_variant(os.path.abspath('tools/_build/libppg_hip_dprof.so'), ('-DPPG_DIRECT_PROFILE',))")
# usage: tools/gpu_pipe_profile.sh TAG [bench args]        (the fused launch: one file, the workgroups' species in word 13)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
export PPG_HIP_LIB=$PWD/tools/_build/libppg_hip_dprof.so PPG_DIRECT_PROFILE_FILE=$PWD/gpurun_out/${tag}_dprof PPG_DIRECT_PROFILE_RUN=${PPG_DIRECT_PROFILE_RUN:-300}
python3 bench.py --workload policy_rollout --steps 20 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/${tag}_dprof.json 2> gpurun_out/${tag}_dprof.err
python3 - <<PY | tee gpurun_out/${tag}_pipe_profile.txt
import numpy as np, json, os
names = {0: "table", 1: "conv3", 2: "barrier wait", 3: "logits+actions", 4: "head", 5: "noise (A) / request (B)", 11: "stage", 6: "private barrier 1", 7: "conv1", 8: "private barrier 2",
         9: "conv2", 10: "barrier wait", 14: "actions"}
print(json.loads(open("gpurun_out/${tag}_dprof.json").readlines()[-1])["roofline"])
files = {}
if os.path.exists("gpurun_out/${tag}_dprof.fused"):
    allw = np.fromfile("gpurun_out/${tag}_dprof.fused", dtype=np.uint64).reshape(-1, 8, 16).astype(np.float64)
    files = {"prey": allw[allw[:, 0, 13] == 2], "pred": allw[allw[:, 0, 13] == 1]}
else:
    files = {sp: np.fromfile("gpurun_out/${tag}_dprof." + sp, dtype=np.uint64).reshape(-1, 8, 16).astype(np.float64) for sp in ("prey", "pred")}
for sp, a in files.items():
    a = a[a[:, 4, 15] > 0]
    its = a[:, 4, 15].mean()
    print(f"== {sp}: {len(a)} workgroups, iterations per workgroup {its:.1f}")
    for role, waves, keys in (("A", range(0, 4), (0, 3, 1, 5, 2)), ("B", range(4, 8), (0, 4, 11, 5, 6, 7, 8, 9, 10))):
        tot = a[:, waves, :13].sum(axis=2) + a[:, waves, 14]
        print(f"  role {role}: cycles per wavefront {tot.mean():.0f} (clock64 ticks); per iteration: " +
              "  ".join(f"{names[k]} {a[:, waves, k].mean() / its:.0f}" for k in keys))
        for w in waves:
            print(f"    wave {w}: " + "  ".join(f"{names[k]} {a[:, w, k].sum() / (a[:, w, :13].sum() + a[:, w, 14].sum()) * 100:.1f}%" for k in keys))
PY
rm -f gpurun_out/${tag}_dprof.prey gpurun_out/${tag}_dprof.pred gpurun_out/${tag}_dprof.fused
